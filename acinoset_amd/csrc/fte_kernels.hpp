// Internal declarations shared by the FTE translation units (not part of the C ABI).
#pragma once
#include <utility>
#include <vector>

#include "common.hpp"

namespace acino {

constexpr int NP = ACINO_N_ACTIVE;   // 25 active states per frame
constexpr int NL = ACINO_N_MARKERS;  // 20 markers
constexpr int BS = ACINO_BS;         // 80 = 3 frames x 25 states + 5 identity pad rows
constexpr int NGRP = 14;             // kinematic frames (rotation groups)
constexpr int FPB = 8;               // frames per workgroup in the assembly kernel
constexpr double FIX_SCALE = 1.1805916207174113e21;  // 2^70: diagonal boost pinning a bound-active variable
// Marquardt scaling lam * diag(H) cannot lift a diagonal entry that is exactly 0: a state no camera observes in a clip of
// fewer than 4 frames (no third-difference row either).  Its damping term is lam * DIAG_FLOOR instead: the variable is
// decoupled (zero row, zero gradient), gets a positive pivot and a step of exactly 0.  Never active otherwise.
constexpr double DIAG_FLOOR = 1e-30;
// Active-set test: a gradient entry below GRAD_ZERO_REL * H_ii (a Newton step of 1e-14 in that variable alone) counts as
// zero.  An entry that vanishes analytically (a joint none of whose markers is detected in the frame) comes out of the
// subtree sums as +-1e-20, and a bare sign test would let that noise decide whether a variable sitting on its bound is
// pinned.  Same rule in oracle.fte.FTEProblem.active_set.
constexpr double GRAD_ZERO_REL = 1e-14;

// Device-resident constant block of one FTE problem.
struct FteConst {
  int32_t n_frames, n_cams;
  int64_t n_global, n_offset;
  int32_t pin_left, pin_right;
  int32_t n_nodes;         // local chain length (super-blocks incl. a pinned left separator)
  int32_t pad;
  double dlc_thresh, inv_r;
  LossC loss;
  double q_w[NP], lo[NP], hi[NP];
  double ftol, xtol, gtol;
  double lam_max;
  int32_t clamp_lambda;
  int32_t precision;       // ACINO_PREC_*
  int64_t clip_len;        // > 0: independent clips of this many frames laid end to end (no coupling across clips)
  int32_t own_lo, own_hi;  // local frames [own_lo, own_hi) count towards cost / pred / step / gradient norms (window sharding)
  double trunc_tol;        // incomplete reduction: largest admissible eps of a dropped coupling (status 7 above it)
  int32_t refine_sweeps;   // block-Jacobi sweeps after the truncated solve: the admissible quantity becomes (2 eps)^(r+1)
  int32_t pad2;
  Cam cams[ACINO_MAX_CAMS];
};

// (D3^T D3)[n, n+k] for global frame n, 0 <= k <= 3, sequence length ng (stencil -1, 3, -3, 1).
__host__ __device__ inline double band_coef(int64_t n, int k, int64_t ng) {
  if (n < 0 || n + k >= ng) return 0.0;
  if (n + k >= 3 && n <= ng - 4) return k == 0 ? 20.0 : (k == 1 ? -15.0 : (k == 2 ? 6.0 : -1.0));   // interior rows
  // (the stencil by selects, not from a table: a table is a load from constant memory, and a kernel that evaluates this
  //  between two prefetches would have to wait for every load in flight)
  auto c = [](int64_t i) { return i == 0 ? -1.0 : (i == 1 ? 3.0 : (i == 2 ? -3.0 : 1.0)); };
  int64_t jlo = n + k - 3 > 0 ? n + k - 3 : 0;
  int64_t jhi = n < ng - 4 ? n : ng - 4;
  double tot = 0.0;
  for (int64_t j = jlo; j <= jhi; ++j) tot += c(n - j) * c(n + k - j);
  return tot;
}

// The same coefficient when the frame axis holds several independent clips of `clip` frames each (0 = one sequence).
__host__ __device__ inline double band_coef_clip(int64_t n, int k, int64_t ng, int64_t clip) {
  if (clip <= 0) return band_coef(n, k, ng);
  if (n < 0 || n + k >= ng) return 0.0;
  const int64_t c0 = n / clip;
  if ((n + k) / clip != c0) return 0.0;
  return band_coef(n - c0 * clip, k, clip);
}

// Per-kernel-class HIP-event profiler (bench.py's live roofline measurement).  Events are recorded on
// the stream the kernels are launched on; nothing is recorded unless enabled.
// one class per kernel: {elim, elim_deep, update0, update, update_deep, backsub0, backsub, trial, assemble, totals,
// control, backsub_tail, trunc_check, chunk_sweep, sep_combine, chunk_backsub, refine}; `units` = chain nodes (BCR kernels) or frames (trial / assemble) the launch processed
enum ProfClass { PC_ELIM = 0, PC_ELIM_DEEP, PC_UPDATE0, PC_UPDATE, PC_UPDATE_DEEP, PC_BACKSUB0, PC_BACKSUB, PC_TRIAL,
                 PC_ASSEMBLE, PC_TOTALS, PC_CONTROL, PC_BACKSUB_TAIL, PC_TRUNC_CHECK, PC_CHUNK_SWEEP, PC_SEP_COMBINE,
                 PC_CHUNK_BACKSUB, PC_REFINE, PC_COUNT };
static_assert(PC_COUNT == ACINO_PROF_CLASSES, "profiler classes");
struct Profiler {
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Span { int cls; size_t a, b; long long units; };
  std::vector<Span> spans;
  hipEvent_t next() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
  ~Profiler() {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  }
};
struct ProfSpan {
  Profiler* p;
  hipStream_t s;
  size_t a = 0;
  int cls;
  long long units;
  ProfSpan(Profiler* prof, int c, hipStream_t st, long long n_units = 0)
      : p(prof && prof->on ? prof : nullptr), s(st), cls(c), units(n_units) {
    if (p) {
      a = p->used;
      hipEvent_t e = p->next();
      if (e) (void)hipEventRecord(e, s); else p = nullptr;
    }
  }
  ~ProfSpan() {
    if (p) {
      size_t b = p->used;
      hipEvent_t e = p->next();
      if (e) {
        (void)hipEventRecord(e, s);
        p->spans.push_back({cls, a, b, units});
      }
    }
  }
};

// x iterate buffers carry 3 halo frames on each side: row (i + 3) holds local frame i.
constexpr int HALO = 3;

// The Gauss-Newton block of a frame is symmetric: the assembly stores its 325 UNORDERED state pairs {p, p'} only, pair
// e = 13 p + d with p' = (p + d) mod 25, d = 0 .. 12 (every cyclic distance once: 25 is odd) - the order in which the chunk
// sweep builds a node from state pairs, so its 256 builder threads read consecutive doubles (round 5; before: [25][25], both
// triangles written, 5 000 B per frame written and read back instead of 2 600).
constexpr int HPAIRS = 13 * NP;
__host__ __device__ inline int hpair(int p, int pc) {
  int d = pc - p;
  if (d < 0) d += NP;
  return d <= NP / 2 ? 13 * p + d : 13 * pc + (NP - d);
}

int launch_assemble(const FteConst* d_c, const FteConst& h_c, const acino_fte_state* d_st, int which,
                    const double* d_det, double* const x[2], double* const H[2], double* const g[2],
                    double* const hd[2], double* d_cost_partials, int* d_nbehind, bool need_jac, bool respect_status, hipStream_t s);
int n_assemble_blocks(int n_frames);
int launch_fk(const double* d_q, int64_t n, double* d_pos, hipStream_t s);
int launch_fk_active(const double* d_xa_halo, int64_t n, double* d_pos, hipStream_t s);

}  // namespace acino
