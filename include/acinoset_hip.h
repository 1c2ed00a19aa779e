/*
 * acinoset_hip.h - C ABI of libacinoset_hip.so (MI355X / gfx950, fp64).
 *
 * Drop-in boundary for the AcinoSet triangulation + Full-Trajectory-Estimation hot path.
 * The reference (pure Python) has no FFI; these are the entry points a binding for that path
 * binds, one per reference interface it replaces (paths relative to the AcinoSet tree):
 *
 *   acino_undistort_fisheye          cv2.fisheye.undistortPoints      src/calib/calib.py:124-125
 *   acino_triangulate_fisheye        triangulate_points_fisheye       src/calib/calib.py:121-130
 *   acino_triangulate_pinhole        triangulate_points               src/calib/calib.py:52-61
 *   acino_project_fisheye            project_points_fisheye           src/calib/calib.py:132-136
 *   acino_project_pinhole            project_points                   src/calib/calib.py:64-66
 *   acino_triangulate_pairs          get_pairwise_3d_points_from_df   src/calib/calib.py:394-423 (triangulate_func = triangulate_points_fisheye)
 *   acino_triangulate_pairs_pinhole  get_pairwise_3d_points_from_df   src/calib/calib.py:394-423 (triangulate_func = triangulate_points; app.py:215-218)
 *   acino_reproject_residuals        project(triangulate(.)) - pts    src/calib/calib.py:312-316 (cost_func_points_only)
 *   acino_triangulate_reproject      the two above fused (one pass over the detections)
 *   acino_cheetah_fk                 pose_to_3d                       src/all_optimizations.py:66-190
 *   acino_fte_*                      the Pyomo model + opt.solve()    src/all_optimizations.py:283-556
 *
 * Conventions
 *   - every function returns 0 on success or a negative acino_status; no C++ exception and no
 *     abort crosses the boundary; acino_last_error_string() describes the last failure (thread-local).
 *   - all pointers named d_* are DEVICE pointers (HBM), caller-allocated and caller-owned; the
 *     library allocates no device memory.  FTE scratch is one caller buffer whose size comes
 *     from acino_fte_workspace_bytes().
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is stream-ordered,
 *     nothing synchronises the device unless stated.
 *   - all arrays are dense, C-contiguous, double (fp64) unless stated.
 */
#ifndef ACINOSET_HIP_H
#define ACINOSET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACINO_ABI_VERSION 3   /* 2: acino_fte_params grew (chunk_nodes, refine_sweeps), 17 profiler classes, status 5-7 / numeric_err bit mask;
                               * 3: acino_skel_fte_* (generic-skeleton FTE), debug-stamp buffer of 72 entries with the selectors at [64], [65] */

typedef enum acino_status {
  ACINO_OK = 0,
  ACINO_ERR_INVALID_ARG = -1,
  ACINO_ERR_HIP = -2,            /* a HIP runtime call or kernel launch failed        */
  ACINO_ERR_WORKSPACE = -3,      /* workspace too small / misaligned                 */
  ACINO_ERR_NO_DEVICE = -4,
  ACINO_ERR_UNSUPPORTED = -5,
  ACINO_ERR_NUMERIC = -6,        /* non-positive pivot in the block factorisation    */
  ACINO_ERR_CALLBACK = -7        /* a caller-supplied reduction callback failed      */
} acino_status;

/* ---- camera records -------------------------------------------------------------------------
 * Fisheye camera: 24 doubles  [fx fy cx cy | k1 k2 k3 k4 | R(9,row-major) | t(3) | alpha 0 0 0]
 * Pinhole camera: 32 doubles  [fx fy cx cy | d(14: k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 tx ty) | R(9) | t(3) | 0 0]
 * (alpha = K[0][1]/fx, OpenCV's skew; zero for every reference rig.) */
#define ACINO_CAM_STRIDE 24
#define ACINO_PINHOLE_STRIDE 32
#define ACINO_MAX_CAMS 16
#define ACINO_MAX_PAIR_CAMS 8   /* dense pair path: the pair mask is one byte (pairs (c, c+1), c < 7) */
#define ACINO_N_MARKERS 20     /* cheetah markers (all_optimizations.py:170-179)            */
#define ACINO_N_STATES 45      /* x,y,z, phi_0..13, theta_0..13, psi_0..13                   */
#define ACINO_N_ACTIVE 25      /* states with Q != 0 (all_optimizations.py:245-252)          */

const char* acino_last_error_string(void);
int acino_abi_version(void);
/* sha256 prefix over every HIP source and header this binary was built from (acinoset_amd/_lib.py::source_hash):
 * build() and the profile stamps compare it instead of file times. */
const char* acino_build_id(void);
/* Number of visible HIP devices, or a negative status. */
int acino_device_count(void);

/* ---- point-wise camera model (a-1, a-2) ------------------------------------------------------ */
/* d_pts[M][2] pixel -> d_out[M][2] normalised coordinates; OpenCV criteria (max_iter=10, eps=1e-8);
 * non-converged / sign-flipped points are written as -1e6 like OpenCV >= 4.5. */
int acino_undistort_fisheye(const double* d_pts, int64_t m, const double* d_cam24, double* d_out,
                            int max_iter, double eps, void* stream);
/* Two-view triangulation: undistort both, 4x4 DLT, smallest right-singular vector, dehomogenise.
 * d_pts1/d_pts2 [M][2], d_cam_a/d_cam_b one camera record each, d_out[M][3]. */
int acino_triangulate_fisheye(const double* d_pts1, const double* d_pts2, int64_t m,
                              const double* d_cam_a24, const double* d_cam_b24, double* d_out, void* stream);
int acino_triangulate_pinhole(const double* d_pts1, const double* d_pts2, int64_t m,
                              const double* d_cam_a32, const double* d_cam_b32, double* d_out, void* stream);
/* d_obj[M][3] -> d_out[M][2]. */
int acino_project_fisheye(const double* d_obj, int64_t m, const double* d_cam24, double* d_out, void* stream);
int acino_project_pinhole(const double* d_obj, int64_t m, const double* d_cam32, double* d_out, void* stream);

/* ---- dense adjacent-pair triangulation (a-3, BASELINE configs[1]) ---------------------------
 * d_det[N][C][L][3] = (x, y, likelihood).  A detection is valid iff likelihood > thresh.
 * For each (frame, marker): triangulate every adjacent camera pair (c, c+1) with both valid, take
 * the Kahan-compensated mean in pair order (what pandas' groupby().mean() computes).
 * d_tri[N][L][3] (NaN when no pair), d_npairs[N][L] u8, d_pairmask[N][L] u8 (bit c = pair (c,c+1)).
 * d_npairs / d_pairmask may be NULL.  1 <= n_cams <= ACINO_MAX_PAIR_CAMS. */
int acino_triangulate_pairs(const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                            const double* d_cams24, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask,
                            void* stream);
/* Initial iterate of the FTE solve from the dense triangulation d_tri[N][n_markers][3] (NaN = no pair), formed on the device
 * (acinoset_amd.fte.triangulation_init; the reference's own initialisation is the nose line, all_optimizations.py:268-277,
 * which acinoset_amd.fte.nose_line_init restates): d_xa[N][n_active] <- 0 except columns 0..2 = mean of the finite ones among
 * markers 0, 1, 2 (eyes, nose) and column psi_column = np.unwrap(atan2) of marker 2 - marker 3 (neck_base -> nose), each
 * np.interp'ed over the frames that lack it (held flat at the ends).  *d_flag (preset to 0 by the caller) <- 1 when no frame
 * has a head marker.  One launch; d_scratch: acino_fte_triangulation_init_scratch_bytes(N), 8-byte aligned. */
size_t acino_fte_triangulation_init_scratch_bytes(int64_t n_frames);
int acino_fte_triangulation_init(const double* d_tri, int64_t n_frames, int n_markers, double* d_xa, int n_active,
                                 int psi_column, void* d_scratch, size_t scratch_bytes, int32_t* d_flag, void* stream);
/* The same index path with the injected pinhole pair (calib.py:52-61): d_cams32 = C pinhole records. */
int acino_triangulate_pairs_pinhole(const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                                    const double* d_cams32, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask,
                                    void* stream);
/* The two calls below in one pass over d_det (BASELINE configs[1]: triangulation + reprojection residual of the
 * triangulated points in every camera): d_tri / d_npairs / d_pairmask as acino_triangulate_pairs, d_res / d_sums as
 * acino_reproject_residuals applied to d_tri. */
int acino_triangulate_reproject(const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                                const double* d_cams24, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask,
                                double* d_res, double* d_sums, void* stream);
/* Reprojection residual of d_pts3[N][L][3] in every camera against d_det:
 * d_res[N][C][L][2] = project(pts3) - det.xy where det valid and pts3 finite, else NaN.
 * d_sums[4] (may be NULL) += {count, sum r, sum r^2, 0.5*sum log1p(r^2)} over valid residual components. */
int acino_reproject_residuals(const double* d_pts3, const double* d_det, int64_t n_frames, int n_cams,
                              int n_markers, double thresh, const double* d_cams24, double* d_res,
                              double* d_sums, void* stream);

/* ---- cheetah forward kinematics (a-4) ---------------------------------------------------------
 * d_q[N][45] full state -> d_pos[N][20][3]. */
int acino_cheetah_fk(const double* d_q, int64_t n_frames, double* d_pos, void* stream);

/* ---- Full Trajectory Estimation (a-5 .. a-11) -------------------------------------------------
 * Reduced problem of the reference NLP: unknowns xa[N][25] (active states),
 *   F = sum rho(w*(pi_c(FK_l(x_n)) - z)) + sum_{n>=3} q_p (x_n - 3x_{n-1} + 3x_{n-2} - x_{n-3})_p^2,
 * box bounds lo/hi, solved by a projected Levenberg-Marquardt whose Gauss-Newton system is
 * block-tridiagonal in super-blocks of 3 frames and solved by block cyclic reduction.
 * Active set: a variable sitting ON a bound whose gradient entry pushes outward is pinned for the iteration (step exactly
 * 0); "pushes" means |g_i| > 1e-14 * H_ii, i.e. a gradient below a Newton step of 1e-14 counts as zero.  Damping is
 * Marquardt's lam * diag(H); a diagonal entry that is exactly 0 (clips of < 4 frames with an unobserved state) is damped
 * by lam * 1e-30 and keeps its variable where it is. */
typedef struct acino_fte_params {
  int32_t n_frames;        /* local frames on this GPU                                      */
  int32_t n_cams;
  int64_t n_global;        /* frames in the whole sequence (== n_frames on one GPU)         */
  int64_t n_offset;        /* global index of local frame 0 (multiple of 3 when > 0)        */
  int32_t pin_left;        /* 1: local chain starts with a separator owned by the left rank */
  int32_t pin_right;       /* 1: the last local super-block is a separator (not eliminated) */
  double dlc_thresh;       /* likelihood threshold (all_optimizations.py:304)               */
  double inv_r_meas;       /* 1/R, R = 5 px (all_optimizations.py:243)                      */
  double redesc_a, redesc_b, redesc_c;   /* 3, 10, 20 (all_optimizations.py:25-27)          */
  double q_w[ACINO_N_ACTIVE];            /* (1/Q_p) / Ts^4 per active state                 */
  double lo[ACINO_N_ACTIVE], hi[ACINO_N_ACTIVE];   /* box bounds (+-inf = free)             */
  double lam0;             /* initial LM damping                                            */
  double ftol, xtol, gtol; /* stopping tolerances (0 disables a test)                       */
  double lam_max;          /* damping ceiling (0 = 1e16)                                    */
  int32_t clamp_lambda;    /* 0: stop with status 4 when lam exceeds lam_max; 1: clamp and keep iterating */
  int32_t shared_gpu;      /* 1: other solver contexts run on this GPU at the same time (batched clips, several ranks on one
                            * device): kernels that spin-wait on other workgroups (the fused back-substitution tail) are
                            * replaced by their per-level forms - concurrent spin-waiting kernels could fill the CUs */
  int64_t clip_len;        /* 0: the frames are ONE sequence.  > 0: the frames are n_frames / clip_len independent clips of
                            * this many frames laid end to end (BASELINE config 5): the smoothness prior does not couple
                            * frames of different clips, everything else - kernels, schedule, one LM controller over the
                            * sum of the clips' costs - is unchanged.  Single-GPU contexts only. */
  int32_t precision;       /* ACINO_PREC_F64 (0, default) or ACINO_PREC_BF16_ROWS (1) = BASELINE config 5's "bf16 residuals
                            * with fp32 accumulate": FK and camera-frame coordinates in fp64, projection / 2x3 Jacobian /
                            * robust weights in fp32, every per-(frame, camera, marker) residual and Jacobian ROW rounded
                            * to bf16 (round-to-nearest-even) before it enters the normal equations, the per-marker
                            * M_l = sum J^T W J and v_l = sum J^T w rho' accumulated in fp32; everything from the 6x6
                            * spatial blocks onward - subtree sums, H, g, the smoothness prior with its 1/Ts^4 weights, the
                            * band factorisation, the LM controller and the cost - stays fp64. */
  int32_t bcr_levels;      /* 0 (default): complete block cyclic reduction.  K > 0: INCOMPLETE reduction - after K levels the
                            * couplings between the remaining nodes (3 * 2^K frames apart) are dropped and every remaining
                            * node is solved on its own.  The dropped blocks decay geometrically with the node distance
                            * (the Gauss-Newton matrix is banded SPD); their normalised size
                            * eps = max || L_b^-1 C L_a^-T ||_F is MEASURED on the device every iteration
                            * (acino_fte_state::trunc_eps) - the solve's relative energy-norm error is <= eps / (1 - eps) -
                            * and an iteration with eps > trunc_tol stops the solve with status 7 instead of returning an
                            * unverified step.  Single-GPU contexts only (no pinned separators). */
  double trunc_tol;        /* admissible eps of an incomplete reduction (0 = 1e-10)                                */
  int32_t own_first;       /* overlapping-window sharding (acinoset_amd/dist.py WindowedFTE): the n_frames of this context */
  int32_t own_count;       /* are a WINDOW [n_offset, n_offset + n_frames) of the sequence, of which only the local frames
                            * [own_first, own_first + own_count) are owned: the whole window is assembled and solved
                            * (frames outside it keep delta = 0), but cost, predicted reduction, step and gradient norms
                            * count owned frames only.  own_count = 0 (default): every frame is owned. */
  int32_t chunk_nodes;     /* linear solver of single-GPU contexts (no pinned separators).  0 (default): chunked
                            * substructuring (csrc/chunk.hip) - the chain of 3-frame nodes is cut into runs of an
                            * automatically chosen length, one workgroup eliminates the interior nodes of a run in order
                            * with every operand in LDS, block cyclic reduction solves the chain of the runs' separators.
                            * m >= 2: the same with m nodes per run (m - 1 interior + 1 separator).  -1: block cyclic
                            * reduction over the whole chain (the round-1/2 solver).  With chunking, bcr_levels counts
                            * reduction levels of the SEPARATOR chain. */
  int32_t refine_sweeps;   /* incomplete reduction (bcr_levels > 0): block-Jacobi sweeps that re-introduce the dropped
                            * couplings after the truncated solve (0 = none).  In theory r sweeps leave a relative
                            * energy-norm error <= (2 eps)^(r+1) / (1 - 2 eps).  What the device CHECKS against trunc_tol with
                            * r > 0 is an a-posteriori estimate from the sweeps themselves (eps is not measured then):
                            * rho / (1 - rho) * max|last update| / max|x| with rho = a ratio of the max-norms of consecutive
                            * updates: of the first two sweeps and - while those are above the rounding level, 2^-44 max|x| -
                            * of the last two, the larger of both (r = 1, or a first update already at rounding: rho = 1/2
                            * assumed); rho > 1/2 refuses the step (status 7).  The estimate is written to
                            * acino_fte_state::trunc_eps. */
} acino_fte_params;
#define ACINO_PREC_F64 0
#define ACINO_PREC_BF16_ROWS 1
#define ACINO_PREC_BF16_RES 2    /* as BF16_ROWS but only the residual rows are rounded to bf16; Jacobian rows stay fp32 */

/* LM state mirrored in device memory (read back with acino_fte_get_state). */
typedef struct acino_fte_state {
  double cost;             /* F at the current iterate (this rank's share when sharded)     */
  double cost_trial;
  double lam, nu;
  double gain, pred, step_inf, gnorm_inf;
  int32_t iter;            /* LM iterations performed                                       */
  int32_t accepted;
  int32_t status;          /* 0 running, 1 ftol, 2 xtol, 3 gtol, 4 lambda overflow, 5 numeric (non-positive pivot),
                            * 6 device synchronisation timeout (backsub tail), 7 dropped couplings above trunc_tol */
  int32_t cur;             /* which of the two iterate buffers is current                   */
  int32_t n_behind;        /* weighted detections with z_cam < 1e-6 (kept, as the reference)  */
  int32_t last_accept;
  int32_t pad0, pad1;
  double trunc_eps;        /* incomplete reduction: what was compared with trunc_tol in the last iteration - the measured
                            * size of the dropped couplings, or (refine_sweeps > 0) the sweeps' error estimate (0: complete) */
} acino_fte_state;

typedef struct acino_fte_ctx acino_fte_ctx;   /* opaque host handle */

/* sizeof() of the two ABI structs as this library was compiled (bindings assert their own layout against it). */
size_t acino_sizeof_fte_params(void);
size_t acino_sizeof_fte_state(void);
size_t acino_fte_workspace_bytes(const acino_fte_params* p);
/* Creates a handle over caller-owned buffers.  d_det[N][C][L=20][3], d_cams24[C][24],
 * d_workspace >= acino_fte_workspace_bytes(p), 256-byte aligned.  Synchronises the stream once. */
int acino_fte_create(acino_fte_ctx** out, const acino_fte_params* p, const double* d_det,
                     const double* d_cams24, void* d_workspace, size_t workspace_bytes, void* stream);
int acino_fte_destroy(acino_fte_ctx* ctx);
/* Layout the linear solver chooses for these parameters: out[0] = nodes per run of the chunked solver (0: block cyclic
 * reduction over the whole chain), out[1] = runs, out[2] = separators, out[3] = reduction levels of the reduced chain
 * (the separators, or the whole chain) when nothing is truncated.  A caller that wants an incomplete reduction picks
 * bcr_levels from the node distance it implies: 3 * max(out[0], 1) * 2^bcr_levels frames. */
int acino_fte_plan(const acino_fte_params* p, int32_t* out);
/* Loads the initial iterate (d_x0[N][25], clipped to the bounds), restarts the LM controller and evaluates
 * cost / gradient / Gauss-Newton blocks.  = load_x + eval(0) + control(NULL, init=1). */
int acino_fte_set_x(acino_fte_ctx* ctx, const double* d_x0, void* stream);
/* One LM iteration, entirely stream-ordered (no host sync): damped block system -> block cyclic reduction ->
 * trial iterate -> residuals, Jacobians, normal-equation assembly at the trial -> accept/reject + lambda. */
int acino_fte_step(acino_fte_ctx* ctx, void* stream);
/* on != 0: acino_fte_step captures its launch sequence into a hipGraph on first use and replays it afterwards
 * (needs a non-null stream; ignored while profiling is active and for sharded contexts). */
int acino_fte_enable_graph(acino_fte_ctx* ctx, int on);
/* Up to max_iter LM iterations (the device stops by itself on convergence; the host peeks every 8 steps);
 * synchronises at the end and fills *out (may be NULL). */
int acino_fte_solve(acino_fte_ctx* ctx, int max_iter, acino_fte_state* out, void* stream);
int acino_fte_get_state(acino_fte_ctx* ctx, acino_fte_state* out, void* stream);   /* synchronises */
/* Current iterate -> d_x[N][25]; positions d_pos[N][20][3]; dx/ddx by the reference's backward-Euler
 * relations (all_optimizations.py:369-383).  Any output may be NULL.  Synchronises. */
int acino_fte_get_result(acino_fte_ctx* ctx, double ts, double* d_x, double* d_pos, double* d_dx, double* d_ddx,
                         void* stream);
/* Switches the arithmetic of the assembly for the following evaluations (ACINO_PREC_*): a mixed-precision solve is
 * finished ("polished") with a few fp64 iterations this way.  Drops the captured step graph; the next
 * acino_fte_step re-evaluates nothing by itself - call acino_fte_reevaluate to refresh cost / gradient / blocks of
 * the current iterate in the new precision before stepping. */
int acino_fte_set_precision(acino_fte_ctx* ctx, int precision);
/* Re-evaluates the CURRENT iterate (cost, gradient, Gauss-Newton blocks) and restarts the controller's stopping
 * state (status -> running; lambda kept, or back to lam0 when the run had ended in lambda overflow). */
int acino_fte_reevaluate(acino_fte_ctx* ctx, void* stream);
/* Copies n frames of the current (which = 0) or trial (which = 1) iterate, starting at local frame `first`
 * (-3 <= first, first + n <= n_frames + 3: the three halo rows on either side are addressable), to d_buf[n][25]
 * (import = 0) or from it (import = 1).  The overlapping-window driver exchanges its edge slabs with these. */
int acino_fte_copy_frames(acino_fte_ctx* ctx, int which, int import, int first, int n, double* d_buf, void* stream);
/* Cost only at d_x[N][25] -> d_cost[1]; evaluated in the trial buffer (call between LM steps). */
int acino_fte_cost(acino_fte_ctx* ctx, const double* d_x, double* d_cost, void* stream);
/* Gradient d_g[N][25] and Gauss-Newton blocks d_h[N][25][25] (measurement part + smoothness diagonal) of the
 * CURRENT iterate, for parity checks.  Either may be NULL. */
int acino_fte_get_grad_hess(acino_fte_ctx* ctx, double* d_g, double* d_h, void* stream);
/* Live per-kernel timing for bench.py: HIP events recorded on the launch stream around every kernel between
 * begin and end.  end synchronises and returns, per class {elim, elim_deep, update0, update, update_deep, backsub0,
 * backsub, trial, assemble, totals, control, backsub_tail, trunc_check, chunk_sweep, sep_combine, chunk_backsub, refine} (one class per kernel), the summed event time in ms, the launch
 * count and the work units (chain nodes for the block-reduction kernels, frames for trial/assemble; may be NULL). */
#define ACINO_PROF_CLASSES 17
int acino_fte_profile_begin(acino_fte_ctx* ctx);
/* Test aid: copies an internal buffer of the linear solver to d_out (at most n doubles).  what: 0 = the solution vector
 * per 3-frame node [n_nodes][80] (after acino_fte_backsub_local); separator-side buffers of the chunked solver: 1 = D,
 * 2 = b, 3 = coupling blocks, 4 = left-run contributions AL (5 is not assigned: ACINO_ERR_INVALID_ARG); 6 = G_k of the
 * interior nodes (lower 16 x 16 tiles), 7 = f_k = F_k x_L, [80] per interior node (T_k is not stored). */
int acino_fte_debug_read(acino_fte_ctx* ctx, int what, double* d_out, int64_t n, void* stream);
/* Debug aid: phase timestamps (wall_clock64 ticks) of ONE workgroup of the elimination kernel / the chunk sweep ->
 * d_dbg[0..63] of a caller buffer of ACINO_DEBUG_STAMP_ENTRIES (72) int64 entries; the caller sets d_dbg[64] = workgroup
 * index and d_dbg[65] = reduction level (k_bcr_elim) or node of the run (k_chunk_sweep) to stamp (read by every launch
 * while enabled).  NULL disables. */
#define ACINO_DEBUG_STAMP_ENTRIES 72
int acino_fte_debug_stamps(acino_fte_ctx* ctx, long long* d_dbg);
int acino_fte_profile_end(acino_fte_ctx* ctx, double* ms_by_class, int* launches_by_class, int64_t* units_by_class,
                          void* stream);
/* Stand-alone helpers: dx/ddx of a trajectory d_x[N][25]; FK of active states d_xa[N][25] -> d_pos[N][20][3]. */
int acino_fte_derivatives(const double* d_x, int64_t n_frames, double ts, double* d_dx, double* d_ddx, void* stream);
int acino_fk_active(const double* d_xa, int64_t n_frames, double* d_pos, void* stream);

/* ---- pieces of one LM iteration, for the multi-GPU driver (acinoset_amd/dist.py) ---------------
 * A sharded sequence gives every rank a contiguous frame block; the last super-block (3 frames) of every
 * rank but the last is a SEPARATOR that the rank does not eliminate (pin_right), and every rank but the
 * first sees its left neighbour's separator as chain node 0 (pin_left).  Per iteration:
 *   reduce_local -> export_separators -> [all-reduce SUM] -> solve_separators -> backsub_local -> trial
 *   -> export_edges(1) -> [all-gather] -> set_halo(1) -> eval(1) -> export_partials -> [all-gather + combine]
 *   -> control(total, 0).
 * `which`: 0 = current iterate, 1 = trial iterate (resolved on the device). */
#define ACINO_BS 80
/* Doubles in the exchange record of ONE separator: D[80][80] | C[80][80] = block(next separator, this one) | b[80]. */
#define ACINO_SEP_DOUBLES (2 * ACINO_BS * ACINO_BS + ACINO_BS)
int acino_fte_load_x(acino_fte_ctx* ctx, const double* d_x0, void* stream);
/* d_halo_l[3][25] = the 3 frames left of the shard, d_halo_r[3][25] the 3 frames right of it; NULL = zeros
 * (sequence end: the coefficients there are zero anyway). */
int acino_fte_set_halo(acino_fte_ctx* ctx, int which, const double* d_halo_l, const double* d_halo_r, void* stream);
/* Residuals + Jacobians + assembly of iterate `which`; local sums {cost, pred, step_inf, gnorm_inf,
 * n_behind, 0, 0, 0} are left in the context (export_partials copies them to d_partial[8]). */
int acino_fte_eval(acino_fte_ctx* ctx, int which, void* stream);
int acino_fte_export_partials(acino_fte_ctx* ctx, double* d_partial, void* stream);
/* Accept/reject + lambda update from d_total[8] (NULL = this context's own sums); init=1 only records the
 * cost of the freshly loaded iterate. */
int acino_fte_control(acino_fte_ctx* ctx, const double* d_total, int init, void* stream);
int acino_fte_reduce_local(acino_fte_ctx* ctx, void* stream);
/* WRITES this rank's contributions into d_sep[world-1][ACINO_SEP_DOUBLES] (caller zero-fills before;
 * contributions of different ranks never overlap except D and b, which the all-reduce sums). */
int acino_fte_export_separators(acino_fte_ctx* ctx, double* d_sep, int rank, int world, void* stream);
size_t acino_sep_scratch_bytes(int n_sep);
/* Solves the all-reduced separator chain (every rank redundantly): d_sep_x[n_sep][80]. */
int acino_solve_separators(const double* d_sep, int n_sep, double* d_sep_x, void* d_scratch, size_t scratch_bytes,
                           void* stream);
int acino_fte_backsub_local(acino_fte_ctx* ctx, const double* d_sep_x, int rank, int world, void* stream);
int acino_fte_trial(acino_fte_ctx* ctx, void* stream);
/* First / last 3 frames of iterate `which` -> d_edge[6][25]. */
int acino_fte_export_edges(acino_fte_ctx* ctx, int which, double* d_edge, void* stream);

/* ---- sparse bundle adjustment ------------------------------------------------------
 * Replaces scipy.optimize.least_squares(method='trf', loss='cauchy', f_scale=...) inside
 * bundle_adjust_points_and_extrinsics (src/calib/calib.py:345-390) and bundle_adjust_points_only
 * (src/calib/calib.py:307-341).  Cost = sum over observations and both pixel axes of 0.5 f^2 log1p((r/f)^2),
 * scipy's definition, so costs compare 1:1 with `res.cost`.  Observations arrive flat: uv[M][2], cam_idx[M]; the host
 * also supplies the CSR grouping by point (pt_start[P+1], pt_obs[M]).  Poses are [R row-major 9 | t 3] per camera and
 * are updated in place together with the points.  A (point, camera) pair may carry at most ONE observation (refused with
 * ACINO_ERR_INVALID_ARG otherwise).  Up to seven cameras take the fused path: one GPU lane per (point, camera) slot, the
 * 6 x 3 coupling blocks never leave the chip, the Schur complement onto the cameras is accumulated on the fp64 matrix
 * cores (workspace: n_points x (n_cams x 4 + 144) B).  More cameras - or the environment variable ACINO_SBA_UNFUSED, an
 * independent cross-check - take the table path: coupling blocks in a dense table [point][camera] (n_points x n_cams x 144 B),
 * Schur complement by LDS atomics. */
typedef struct acino_sba_params {
  int32_t n_cams;
  int32_t optimize_cameras;   /* 0 = points only (calib.py:327), 1 = points + extrinsics (calib.py:369) */
  int64_t n_points;
  int64_t n_obs;
  double f_scale;             /* Cauchy scale in px: 1 (scipy default, calib.py:381) or 50 (calib.py:327,335) */
  double lam0;                /* initial Marquardt damping (1e-3) */
  double ftol;                /* stop when an accepted step lowers the cost by <= ftol * cost */
  double gtol;                /* stop when ||J^T W r||_inf <= gtol */
  int32_t max_iter;
  int32_t camera_model;       /* 0 = cv2.fisheye (app.py:220-223), 1 = cv2.projectPoints pinhole (app.py:215-218) */
  int32_t precision;          /* ACINO_PREC_F64 (0) or ACINO_PREC_BF16_ROWS (1): BASELINE config 5's "bf16 residuals with fp32
                               * accumulate" for the extrinsic refinement - residuals and Jacobian rows rounded to bf16, the
                               * point / coupling / camera blocks accumulated in fp32; cost, Schur complement, camera solve
                               * and updates fp64 */
  int32_t pad0;
} acino_sba_params;
typedef struct acino_sba_info {
  double cost_initial, cost_final, gnorm_inf, lam;
  int32_t iterations, accepted;
  int32_t status;             /* 0 max_iter, 1 ftol, 3 gtol, 4 lambda overflow, 5 numeric failure */
  int32_t pad0;
} acino_sba_info;
size_t acino_sizeof_sba_params(void);
size_t acino_sizeof_sba_info(void);
size_t acino_sba_workspace_bytes(int n_cams, int64_t n_points, int64_t n_obs);
/* d_intr[C][16] = fx fy cx cy | 12 distortion coefficients (fisheye: k1..k4, rest 0; pinhole: k1 k2 p1 p2 k3 k4 k5 k6
 * s1..s4); d_res_before / d_res_after [M][2] (projected - observed, as calib.py:316,359) or NULL. */
int acino_sba_solve(const acino_sba_params* prm, const double* d_intr, double* d_Rt, double* d_pts,
                    const double* d_uv, const int32_t* d_cam_idx, const int32_t* d_pt_start, const int32_t* d_pt_obs,
                    void* d_ws, size_t ws_bytes, double* d_res_before, double* d_res_after, acino_sba_info* info,
                    void* stream);
/* The same solve with the POINTS sharded over several processes / GPUs and the cameras replicated (BASELINE config 5,
 * "the SBA extrinsic refinement adds a 36x36 camera-block + 36-vector all-reduce per
 * iteration").  Every rank passes its own points and observations and identical camera poses; `reduce` must combine
 * n doubles at d_buf (device memory inside d_ws) over all ranks in place - op 0: sum, op 1: max - and return 0; it is
 * called after the stream has been synchronised, once at entry (max of the input-check flag: a rank with a duplicate (point,
 * camera) pair or a camera index out of range makes EVERY rank return ACINO_ERR_INVALID_ARG together) and five times per LM
 * iteration (max point gradient; camera blocks + camera
 * gradient, 27 n_cams doubles; the Schur complement and its right-hand side, (6 n_cams)^2 + 6 n_cams; the predicted
 * reduction; the trial cost - and the initial cost once).  All ranks take identical
 * decisions and leave with identical poses.  reduce == NULL is acino_sba_solve. */
typedef int (*acino_reduce_fn)(void* user, double* d_buf, int64_t n, int op, void* stream);
int acino_sba_solve_sharded(const acino_sba_params* prm, const double* d_intr, double* d_Rt, double* d_pts,
                            const double* d_uv, const int32_t* d_cam_idx, const int32_t* d_pt_start,
                            const int32_t* d_pt_obs, void* d_ws, size_t ws_bytes, double* d_res_before,
                            double* d_res_after, acino_sba_info* info, acino_reduce_fn reduce, void* reduce_user,
                            void* stream);

/* ---- generic-skeleton forward kinematics (src/build.py:28-86) ---------------------------
 * The host compiles a skeleton dictionary into <= ACINO_SKEL_MAX_OPS link operations, evaluated in order for every
 * frame:  pose[child] = pose[parent] + M @ off,  M = R_loc or R_loc^T of the PARENT part's own angles
 * (phi, theta, psi at q[3+angle], q[3+L+angle], q[3+2L+angle]; R_loc = Rz(psi) Rx(phi) Ry(theta) restricted to the
 * dofs in flags bits 0..2; bit 3 set = use R_loc, clear = R_loc^T).  All n_pose slots start at the root (x, y, z). */
#define ACINO_SKEL_MAX_OPS 64
typedef struct acino_skel_op {
  int32_t child, parent, angle, flags;
  double off[3];
} acino_skel_op;
/* d_q[N][3 + 3 L] -> d_pos[N][n_pose][3]; h_ops is a HOST array. */
int acino_skeleton_fk(const double* d_q, int64_t n_frames, int n_angles, int n_pose, const acino_skel_op* h_ops,
                      int n_ops, double* d_pos, void* stream);

/* ---- generic-skeleton Full Trajectory Estimation (src/build.py:28-335: build_model + solve_optimisation) -------------
 * The skeleton-driven NLP of the reference in reduced form (oracle/skel_fte.py):
 *   min  sum |w_ncl (pi_c(pose_l(x_n))_d - z_ncld)|  +  sum_{n>=3,p} (model_weight / h^4) (x_n - 3 x_n-1 + 3 x_n-2 - x_n-3)_p^2
 *   s.t. lo[n][p] <= x[n][p] <= hi[n][p]
 * over the n_active states that move a pose (x, y, z and the enabled angles of parent parts; h_active[n_active] holds
 * their indices in the full state [x y z | phi | theta | psi], increasing, starting 0, 1, 2 - every other state stays at
 * its initial 0 as in the reference).  h_ops is the link program of acino_skeleton_fk.  d_meas[N][C][n_pose][2] and
 * d_w[N][C][n_pose] are indexed by POSE SLOT: the caller pairs slots with detections (the reference pairs by position in
 * the skeleton's marker list, build.py:113-128,288-292) and sets w = 1/R where likelihood > threshold, else 0.
 * d_x[N][n_active]: initial iterate in, solution out.  d_pos[N][n_pose][3] (may be NULL): poses of the solution.
 * Solved by the projected Levenberg-Marquardt of the cheetah path (L1 loss: IRLS curvature w^2 / max(|e|, l1_eps), cost
 * and gradient those of |e|); the controller runs on the device (one status word per clip read back per iteration).
 * Limits: n_active <= 64, 2 n_pose C <= 256. */
typedef struct acino_skel_fte_params {
  int32_t n_frames, n_cams, n_pose, n_ops, n_angles, n_active;
  int32_t max_iter, pad0;
  double h;                   /* time step (build.py:131: 1/120)                           */
  double model_weight;        /* build.py:186-191: 0.002 for every state                    */
  double l1_eps;              /* floor of |e| in the IRLS weight (scaled residual units)     */
  double lam0, ftol, xtol, gtol, lam_max;
} acino_skel_fte_params;
typedef struct acino_skel_fte_info {
  double cost_initial, cost_final, gnorm_inf, lam;
  int32_t iterations, accepted;
  int32_t status;             /* 0 max_iter, 1 ftol, 2 xtol, 3 gtol, 4 lambda overflow, 5 numeric failure */
  int32_t pad0;
} acino_skel_fte_info;
size_t acino_sizeof_skel_fte_params(void);
size_t acino_sizeof_skel_fte_info(void);
size_t acino_skel_fte_workspace_bytes(const acino_skel_fte_params* p);
int acino_skel_fte_solve(const acino_skel_fte_params* p, const acino_skel_op* h_ops, const int32_t* h_active,
                         const double* d_meas, const double* d_w, const double* d_cams24, const double* d_lo,
                         const double* d_hi, double* d_x, double* d_pos, void* d_workspace, size_t workspace_bytes,
                         acino_skel_fte_info* info, void* stream);
/* The same for n_clips independent clips of n_frames frames each in one call (same skeleton, same cameras; the reference
 * solves windows of N = 100 frames, build.py:131-133 - a video is many of them): every array gains a leading clip index
 * (d_meas[n_clips][N][C][n_pose][2], d_x[n_clips][N][n_active], ..., infos[n_clips]); no coupling across clips; one
 * workgroup per clip walks its banded factorisation, every clip has its own Levenberg-Marquardt controller on the device
 * and stops on its own criteria.  With n_clips > 1 and infos given, a clip that fails numerically (infos[b].status = 5) does
 * NOT fail the call: the other clips' results stand and the return value is ACINO_OK; the caller reads the status per clip.
 * (n_clips = 1, or no infos: ACINO_ERR_NUMERIC as acino_skel_fte_solve.) */
size_t acino_skel_fte_workspace_bytes_batch(const acino_skel_fte_params* p, int n_clips);
int acino_skel_fte_solve_batch(const acino_skel_fte_params* p, int n_clips, const acino_skel_op* h_ops, const int32_t* h_active,
                               const double* d_meas, const double* d_w, const double* d_cams24, const double* d_lo,
                               const double* d_hi, double* d_x, double* d_pos, void* d_workspace, size_t workspace_bytes,
                               acino_skel_fte_info* infos, void* stream);

/* ---- extended Kalman filter + RTS smoother (src/all_optimizations.py:569-865) ---------
 * One call filters and smooths n_seq independent sequences of n_frames frames (same rig).  States are the reference's
 * 75 = 3 x 25 [pose | velocity | acceleration], pose parameters in the order of qb_list (:734-746).  d_det is
 * [n_seq][n_frames][n_cams][20][3] (x, y, likelihood), d_states0 [n_seq][75] the state BEFORE the first prediction
 * (:700-711), d_est / d_smooth [n_seq][n_frames][75] the filtered and the smoothed states (:848-856 slices them into
 * x, dx, ddx), d_outliers [n_seq] the gated pixel pairs (:818).  Model constants (P0, Q, R, the 3-sigma gate, the
 * forward-difference step 1e-3) are the reference's literals.  The predicted covariances are not stored: the smoother
 * rebuilds P_pred[i+1] = F P_est[i] F^T + Q from the filtered one with the filter's own arithmetic. */
typedef struct acino_ekf_params {
  int64_t n_frames;
  int32_t n_seq;
  int32_t n_cams;             /* <= 6 */
  double fps;
  double dlc_thresh;          /* likelihood < thresh -> measurement sigma = cam_width (:805-808) */
  double cam_width;           /* camera_resolution[0], the reference's max_pixel_err (:611) */
  int32_t smoother_pivoting;  /* 0: Cholesky of P_pred, Gauss-Jordan with partial pivoting only where a pivot fails;
                                 1: always the pivoting solver (the reference's np.linalg.inv makes no definiteness
                                 assumption, :840) */
  int32_t reserved;
} acino_ekf_params;
size_t acino_sizeof_ekf_params(void);
size_t acino_ekf_workspace_bytes(int64_t n_frames, int n_seq);
int acino_ekf_run(const acino_ekf_params* prm, const double* d_det, const double* d_cams24, const double* d_states0,
                  void* d_ws, size_t ws_bytes, double* d_est, double* d_smooth, int32_t* d_outliers, void* stream);

/* The same sharded iteration as FOUR fused phases with the three collectives between them; each phase is a fixed
 * launch sequence on caller-owned buffers and is captured into a hipGraph (acino_fte_enable_graph) per buffer set:
 *   reduce  : zero d_sep, local reduction, export separators            -> all_reduce(d_sep)
 *   solve   : separator solve, local back-substitution, trial iterate,
 *             export its 3+3 edge frames to d_edge_out[6][25]           -> all_gather -> d_all_edges[world][6][25]
 *   eval    : halo from the neighbours' rows of d_all_edges, residuals + Jacobians + assembly of iterate `which`,
 *             local sums to d_partial_out[8]                            -> all_gather -> d_all_partials[world][8]
 *   control : sums combined in rank order, accept/reject + lambda update (init=1: record the loaded iterate). */
/* Bit mask of instantiated graphs: bits 0..3 = the four sharded phases, bit 4 = the whole single-shard step. */
int acino_fte_graphs_active(acino_fte_ctx* ctx);
int acino_fte_shard_reduce(acino_fte_ctx* ctx, double* d_sep, int rank, int world, void* stream);
int acino_fte_shard_solve(acino_fte_ctx* ctx, const double* d_sep, double* d_sep_x, void* d_scratch, size_t scratch_bytes,
                          double* d_edge_out, int rank, int world, void* stream);
int acino_fte_shard_eval(acino_fte_ctx* ctx, int which, const double* d_all_edges, int rank, int world,
                         double* d_partial_out, void* stream);
int acino_fte_shard_control(acino_fte_ctx* ctx, const double* d_all_partials, int world, int init, void* stream);

/* Self-test of the fp64 MFMA tile layout used by the block solver: d_a[16][K], d_b[K][16] -> d_c[16][16]. */
/* Debug aid: n_blocks workgroups that fill 64 KB of LDS each with NaN (to be run on a second stream beside a solve:
 * any kernel that reads LDS it has not written itself then produces NaN). */
int acino_debug_poison_lds(int n_blocks, int spin, void* stream);
int acino_selftest_mfma(const double* d_a, const double* d_b, int k, double* d_c, void* stream);
/* Host only (tests): the table by which the fused narrow levels of the separator reduction (csrc/seplevel.hip) split an
 * eliminated node's work over T workgroups, 1 <= T <= 16.  out[T][64] ints per workgroup: bit mask of the strips of
 * [W_l | W_r] it computes (bit s < 5: columns 16 s .. of W_l, bit 5 + s: of W_r), bit mask of the strips it stores, number
 * of product tiles, then the tile codes (0 .. 24 P_l(a, b) = 5 a + b with a >= b; 25 .. 49 P_r; 50 .. 74 X(a, b)). */
int acino_debug_level_split(int T, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ACINOSET_HIP_H */
