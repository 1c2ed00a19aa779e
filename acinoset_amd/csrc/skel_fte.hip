// Full Trajectory Estimation for a GENERIC skeleton (gfx950, fp64): the reference's skeleton-driven model
// build_model / solve_optimisation of src/build.py:28-335 - pickled skeleton -> sympy poses -> Pyomo NLP -> IPOPT - in the
// reduced form of oracle/skel_fte.py:
//     min_x  sum |w_ncl (pi_c(pose_l(x_n))_d - z_ncld)|  +  sum_{n >= 3, p} (w_model / h^4) (third difference of x_p)^2
//     s.t.   lo[n, p] <= x[n, p] <= hi[n, p]
// (L1 measurement loss build.py:299, constant model weight 0.002 :186-191, limits :263-266), solved by the projected
// Levenberg-Marquardt of the cheetah path with IRLS curvature w^2 / max(|e|, l1_eps).  The cheetah kernels are specialised to
// 25 states and 80-wide three-frame nodes; a skeleton file gives 3 + 3 L states per frame (48 for the shipped human, 36 of
// them move a pose), so this path has its own, size-generic kernels:
//   k_skel_assemble  one workgroup per frame: link program (pose[child] = pose[parent] + M(parent's own angles) off, with
//                    dM/dangle off beside it), fisheye projection + 2x3 Jacobian per (pose, camera), the residual
//                    Jacobian rows A[r][p] in LDS, H_n = A^T W A, g_n, cost
//   k_skel_build     the damped block-banded system: P x P blocks (P padded to a multiple of 16), three sub-diagonals of
//                    diagonal smoothness couplings, bound-active variables deleted (unit row / column)
//   k_skel_solve     ONE workgroup: banded block Cholesky frame by frame - the 4P x P panel of a frame in LDS, diagonal
//                    16 x 16 tiles by the register-resident pivot chain of dense80.hpp, panel / trailing / window updates
//                    as fp64 MFMA tile products - with the forward substitution folded in, then the backward substitution
//   k_skel_trial, k_skel_reduce   trial iterate, predicted reduction, the sums the controller needs
//   k_skel_control   the Levenberg-Marquardt controller, one workgroup per clip (the sums above, accept / reject, damping)
// A call solves n_clips independent clips of n_frames frames each (same skeleton, same cameras): every kernel but the solve
// runs over all frames of all clips, the solve is one workgroup PER CLIP, every clip has its own controller state on the
// device (the host reads the clips' status words back once per iteration and stops when all are done; finished clips are
// skipped by the kernels).  One clip is latency-bound by construction (34 us per frame and iteration in the banded
// factorisation); the reference's own use is windows of 100 frames (build.py:131-133), and a video is many of them.
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <vector>

#include "dense80.hpp"

namespace acino {

constexpr int SK_MAXP = 64;        // active states per frame
constexpr int SK_MAXROWS = 256;    // residual rows per frame = 2 * poses * cameras

struct SkelDev {                   // device-resident description of one problem
  int32_t n_frames, n_cams, n_pose, n_ops, n_act, PT, n_rows, pad;
  double q;                        // model weight / h^4
  double l1_eps, lam_floor;
  acino_skel_op op[ACINO_SKEL_MAX_OPS];
  int8_t amap[ACINO_SKEL_MAX_OPS][4];              // per op: active index of the parent's phi, theta, psi (-1: none)
  unsigned long long pmask[ACINO_SKEL_MAX_OPS + 1];  // per pose slot: the ops on its path from the root
  Cam cams[ACINO_MAX_CAMS];
};

struct SkelClip {                  // controller state of one clip (device)
  double F, lam, nu, gnorm, cost0, Ft, pred, step;
  int32_t cur, status, it, accepted, pivot_err, pad;
};

// ---- assembly ---------------------------------------------------------------------------------------------------
// (frame index n runs over the frames of all clips; `which` = 0: the clips' current iterate, 1: their trial iterate)
template <bool JAC>
__global__ void __launch_bounds__(256)
k_skel_assemble(const SkelDev* __restrict__ dev, const SkelClip* __restrict__ clip, int which, const double* __restrict__ x0,
                const double* __restrict__ x1, const double* __restrict__ meas, const double* __restrict__ wgt,
                double* __restrict__ H0, double* __restrict__ H1, double* __restrict__ g0, double* __restrict__ g1,
                double* __restrict__ hd0, double* __restrict__ hd1, double* __restrict__ c0, double* __restrict__ c1) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const SkelDev& D = *dev;
  const int tid = threadIdx.x, n = blockIdx.x;
  const SkelClip& cs = clip[n / D.n_frames];
  if (cs.status != 0) return;                               // (the clip is finished)
  const int buf = cs.cur ^ which;
  const double* __restrict__ x = buf ? x1 : x0;
  double* __restrict__ H = buf ? H1 : H0;
  double* __restrict__ g = buf ? g1 : g0;
  double* __restrict__ hd = buf ? hd1 : hd0;
  double* __restrict__ cost_part = buf ? c1 : c0;
  const int P = D.n_act, C = D.n_cams, NPOSE = D.n_pose, NOPS = D.n_ops, R = D.n_rows, lda = P | 1;
  double* xs = reinterpret_cast<double*>(smem_raw);          // [64] active states of this frame
  double* opv = xs + SK_MAXP;                                 // [n_ops][4][3]: M off, dM/dphi off, dM/dtheta off, dM/dpsi off
  double* pos = opv + ACINO_SKEL_MAX_OPS * 12;                // [n_pose][3]
  double* jrow = pos + (ACINO_SKEL_MAX_OPS + 1) * 3;          // [R][3] projection Jacobian row
  double* gsr = jrow + SK_MAXROWS * 3;                        // [R] d cost / d residual
  double* hwr = gsr + SK_MAXROWS;                             // [R] IRLS curvature weight
  double* red = hwr + SK_MAXROWS;                             // [8]
  double* A = red + 8;                                        // [R][lda]
  if (tid < P) xs[tid] = x[(size_t)n * P + tid];
  __syncthreads();
  // ---- link operators: R_loc = Rz(psi) Rx(phi) Ry(theta) of the parent's own angles (reference sign convention)
  if (tid < NOPS) {
    const acino_skel_op& o = D.op[tid];
    const int f = o.flags;
    double sp = 0, cp = 1, st = 0, ct = 1, sz = 0, cz = 1;
    if (f & 1) sincos(D.amap[tid][0] >= 0 ? xs[D.amap[tid][0]] : 0.0, &sp, &cp);
    if (f & 2) sincos(D.amap[tid][1] >= 0 ? xs[D.amap[tid][1]] : 0.0, &st, &ct);
    if (f & 4) sincos(D.amap[tid][2] >= 0 ? xs[D.amap[tid][2]] : 0.0, &sz, &cz);
    const double Ry[3][3] = {{ct, 0, -st}, {0, 1, 0}, {st, 0, ct}}, dRy[3][3] = {{-st, 0, -ct}, {0, 0, 0}, {ct, 0, -st}};
    const double Rx[3][3] = {{1, 0, 0}, {0, cp, sp}, {0, -sp, cp}}, dRx[3][3] = {{0, 0, 0}, {0, -sp, cp}, {0, -cp, -sp}};
    const double Rz[3][3] = {{cz, sz, 0}, {-sz, cz, 0}, {0, 0, 1}}, dRz[3][3] = {{-sz, cz, 0}, {-cz, -sz, 0}, {0, 0, 0}};
    auto mul = [](const double (&a)[3][3], const double (&b)[3][3], double (&c)[3][3]) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
    };
    double RxRy[3][3], M[4][3][3], T1[3][3], T2[3][3];
    mul(Rx, Ry, RxRy);
    mul(Rz, RxRy, M[0]);                 // R_loc
    mul(dRx, Ry, T1);
    mul(Rz, T1, M[1]);                   // d / d phi
    mul(Rx, dRy, T1);
    mul(Rz, T1, M[2]);                   // d / d theta
    mul(dRz, RxRy, M[3]);                // d / d psi
    (void)T2;
    const bool untr = (f & 8) != 0;      // bit 3: R_loc itself, else its transpose
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const bool on = m == 0 || ((f >> (m - 1)) & 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double v = 0.0;
        if (on) {
          v = untr ? M[m][i][0] * o.off[0] + M[m][i][1] * o.off[1] + M[m][i][2] * o.off[2]
                   : M[m][0][i] * o.off[0] + M[m][1][i] * o.off[1] + M[m][2][i] * o.off[2];
        }
        opv[(tid * 4 + m) * 3 + i] = v;
      }
    }
  }
  __syncthreads();
  if (tid < 3) {                         // poses, coordinate by coordinate, in program order
    for (int s = 0; s < NPOSE; ++s) pos[s * 3 + tid] = xs[tid];
    for (int k = 0; k < NOPS; ++k) pos[D.op[k].child * 3 + tid] = pos[D.op[k].parent * 3 + tid] + opv[(k * 4) * 3 + tid];
  }
  __syncthreads();
  double my_cost = 0.0;
  // ---- projection of every (pose, camera): residuals, L1 cost, Jacobian rows (pt3d_to_2d, build.py:457-481)
  if (tid < NPOSE * C) {
    const int l = tid / C, c = tid % C;
    const Cam& cam = D.cams[c];
    const double px = pos[l * 3], py = pos[l * 3 + 1], pz = pos[l * 3 + 2];
    const double* mz = meas + (((size_t)n * C + c) * NPOSE + l) * 2;
    const double um = mz[0], vm = mz[1];
    double w = wgt[((size_t)n * C + c) * NPOSE + l];
    if (!(m_finite(um) && m_finite(vm))) w = 0.0;
    const double xc = cam.R[0] * px + cam.R[1] * py + cam.R[2] * pz + cam.t[0];
    const double yc = cam.R[3] * px + cam.R[4] * py + cam.R[5] * pz + cam.t[1];
    const double zc = cam.R[6] * px + cam.R[7] * py + cam.R[8] * pz + cam.t[2];
    if (fabs(zc) < 1e-9) w = 0.0;        // (the singular plane itself; no other cut, as the reference)
    const int r0 = 2 * tid;
    double ju[3] = {0, 0, 0}, jv[3] = {0, 0, 0}, gu = 0, gv = 0, hu = 0, hv = 0;
    if (w != 0.0) {
      const double iz = 1.0 / zc;
      const double a = xc * iz, b = yc * iz;
      const double r2 = a * a + b * b + 1e-12;
      const double r = sqrt(r2), ir = 1.0 / r;
      const double th = atan(r), th2 = th * th;
      const double poly = 1 + th2 * (cam.k1 + th2 * (cam.k2 + th2 * (cam.k3 + th2 * cam.k4)));
      const double thD = th * poly, m = thD * ir;
      const double eu = w * (cam.fx * a * m + cam.cx - um), ev = w * (cam.fy * b * m + cam.cy - vm);
      my_cost = fabs(eu) + fabs(ev);
      if (JAC) {
        const double dthD = 1 + th2 * (3 * cam.k1 + th2 * (5 * cam.k2 + th2 * (7 * cam.k3 + th2 * 9 * cam.k4)));
        const double dm_dr = (dthD / (1 + r2) * r - thD) * (ir * ir);
        const double dm_da = dm_dr * a * ir, dm_db = dm_dr * b * ir;
        const double du_da = cam.fx * (m + a * dm_da), du_db = cam.fx * a * dm_db;
        const double dv_da = cam.fy * b * dm_da, dv_db = cam.fy * (m + b * dm_db);
        const double uc0 = du_da * iz, uc1 = du_db * iz, uc2 = -(du_da * a + du_db * b) * iz;
        const double vc0 = dv_da * iz, vc1 = dv_db * iz, vc2 = -(dv_da * a + dv_db * b) * iz;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          ju[j] = uc0 * cam.R[j] + uc1 * cam.R[3 + j] + uc2 * cam.R[6 + j];
          jv[j] = vc0 * cam.R[j] + vc1 * cam.R[3 + j] + vc2 * cam.R[6 + j];
        }
        gu = w * (eu > 0 ? 1.0 : (eu < 0 ? -1.0 : 0.0));
        gv = w * (ev > 0 ? 1.0 : (ev < 0 ? -1.0 : 0.0));
        hu = w * w / fmax(fabs(eu), D.l1_eps);
        hv = w * w / fmax(fabs(ev), D.l1_eps);
      }
    }
    if (JAC) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        jrow[r0 * 3 + j] = ju[j];
        jrow[(r0 + 1) * 3 + j] = jv[j];
      }
      gsr[r0] = gu;
      gsr[r0 + 1] = gv;
      hwr[r0] = hu;
      hwr[r0 + 1] = hv;
    }
  }
  if (JAC) {
    for (int e = tid; e < R * lda; e += 256) A[e] = 0.0;
    __syncthreads();
    // residual Jacobian: root columns, then one entry per (row, op on the pose's path, enabled angle of the op's parent)
    for (int e = tid; e < R * 3; e += 256) A[(e / 3) * lda + e % 3] = jrow[e];
    for (int e = tid; e < R * NOPS; e += 256) {
      const int r = e / NOPS, k = e % NOPS, l = r / (2 * C);
      if ((D.pmask[l] >> k) & 1ull) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const int p = D.amap[k][ax];
          if (p >= 0) {
            const double* dv = opv + (k * 4 + 1 + ax) * 3;
            A[r * lda + p] = jrow[r * 3] * dv[0] + jrow[r * 3 + 1] * dv[1] + jrow[r * 3 + 2] * dv[2];
          }
        }
      }
    }
    __syncthreads();
    // H_n = A^T W A (upper pairs, mirrored), g_n = A^T gs (+ the smoothness terms, as the cheetah assembly)
    const double b0 = band_coef(n % D.n_frames, 0, D.n_frames);      // (position inside the clip)
    for (int e = tid; e < P * P; e += 256) {
      const int p = e / P, pc = e % P;
      if (pc < p) continue;
      double s = 0.0;
      for (int r = 0; r < R; ++r) s += hwr[r] * A[r * lda + p] * A[r * lda + pc];
      if (p == pc) {
        s += 2.0 * D.q * b0;
        hd[(size_t)n * P + p] = s;
      }
      H[((size_t)n * P + p) * P + pc] = s;
      H[((size_t)n * P + pc) * P + p] = s;
    }
  }
  if (tid < P) {
    const double* xc = x + (size_t)n * P + tid;
    const int nl = n % D.n_frames;                            // position inside the clip: no coupling across clips
    if (nl >= 3) {
      const double d3 = xc[0] - 3.0 * xc[-P] + 3.0 * xc[-2 * P] - xc[-3 * P];
      my_cost += D.q * d3 * d3;
    }
    if (JAC) {
      double gs = 0.0;
#pragma unroll
      for (int k = -3; k <= 3; ++k) {
        const int nn = nl + k;
        if (nn < 0 || nn >= D.n_frames) continue;
        const double bc = k >= 0 ? band_coef(nl, k, D.n_frames) : band_coef(nn, -k, D.n_frames);
        gs += bc * xc[k * P];
      }
      double gm = 0.0;
      for (int r = 0; r < R; ++r) gm += gsr[r] * A[r * lda + tid];
      g[(size_t)n * P + tid] = gm + 2.0 * D.q * gs;
    }
  }
  for (int off = 32; off > 0; off >>= 1) my_cost += __shfl_down(my_cost, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = my_cost;
  __syncthreads();
  if (tid == 0) cost_part[n] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- the damped system ----------------------------------------------------------------------------------------------
// band[n][j] (j = 0..3): block (n + j, n), [PT][PT] row-major; rhs[n][PT]; gn_part[n] = max |projected gradient|.
__device__ __forceinline__ bool skel_fixed(double xv, double gv, double d0, double lo, double hi) {
  const double gtol = GRAD_ZERO_REL * d0;
  return (xv <= lo && gv > gtol) || (xv >= hi && gv < -gtol);
}
__global__ void __launch_bounds__(256)
k_skel_build(const SkelDev* __restrict__ dev, const SkelClip* __restrict__ clip, const double* __restrict__ x0,
             const double* __restrict__ x1, const double* __restrict__ g0, const double* __restrict__ g1,
             const double* __restrict__ H0, const double* __restrict__ H1, const double* __restrict__ hd0,
             const double* __restrict__ hd1, const double* __restrict__ lo, const double* __restrict__ hi,
             double* __restrict__ band, double* __restrict__ rhs, double* __restrict__ gn_part, int final_pass) {
  const SkelDev& D = *dev;
  const int tid = threadIdx.x, n = blockIdx.x, P = D.n_act, PT = D.PT, N = D.n_frames, nl = n % N;
  const SkelClip& cs = clip[n / N];
  if (cs.status != 0 && !final_pass) return;                // (finished; the final pass only wants the gradient norms)
  const double* __restrict__ x = cs.cur ? x1 : x0;
  const double* __restrict__ g = cs.cur ? g1 : g0;
  const double* __restrict__ H = cs.cur ? H1 : H0;
  const double* __restrict__ hd = cs.cur ? hd1 : hd0;
  const double lam = cs.lam;
  __shared__ unsigned char fx[4][SK_MAXP];
  __shared__ double red[4];
  for (int e = tid; e < 4 * P; e += 256) {
    const int j = e / P, p = e % P;
    bool f = false;
    if (nl + j < N) {
      const size_t q = (size_t)(n + j) * P + p;
      f = skel_fixed(x[q], g[q], hd[q], lo[q], hi[q]);
    }
    fx[j][p] = f ? 1 : 0;
  }
  __syncthreads();
  double* B = band + (size_t)n * 4 * PT * PT;
  for (int e = tid; e < PT * PT; e += 256) {
    const int p = e / PT, pc = e % PT;
    double v = 0.0;
    if (p < P && pc < P) {
      if (fx[0][p] || fx[0][pc]) v = p == pc ? 1.0 : 0.0;
      else {
        v = H[((size_t)n * P + p) * P + pc];
        if (p == pc) v += lam * fmax(hd[(size_t)n * P + p], D.lam_floor);
      }
    } else if (p == pc) v = 1.0;
    B[e] = v;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      double c = 0.0;
      if (p == pc && p < P && nl + j < N && !fx[0][p] && !fx[j][p]) c = 2.0 * D.q * band_coef(nl, j, N);
      B[(size_t)j * PT * PT + e] = c;
    }
  }
  double gmax = 0.0;
  if (tid < PT) {
    double b = 0.0;
    if (tid < P && !fx[0][tid]) b = -g[(size_t)n * P + tid];
    rhs[(size_t)n * PT + tid] = b;
    gmax = fabs(b);
  }
  for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_down(gmax, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = gmax;
  __syncthreads();
  if (tid == 0) gn_part[n] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ---- banded block Cholesky + substitutions, one workgroup ----------------------------------------------------------------
// Frame n: panel = [A_nn; A_n+1,n; A_n+2,n; A_n+3,n] (4 PT x PT) in LDS, leading dimension PT + 1.
//   for kb:  diagonal tile -> U_kk = L_kk^-T (register pivot chain, dense80.hpp)
//            tile(t, kb) <- tile(t, kb) U_kk for every row tile below;   tile(rt, ct) -= tile(rt, kb) tile(ct, kb)^T
//   forward: y_n = L_nn^-1 r_n,  r_n+j -= L_n+j,n y_n
//   window : A_n+i,n+j -= L_n+i,n L_n+j,n^T  (1 <= j <= i <= 3; read-modify-write of the band in memory)
// then right to left:  x_n = L_nn^-T (y_n - sum_j L_n+j,n^T x_n+j).  Diagonal tiles keep U_kk, all other tiles L.
constexpr int SK_ST = 512, SK_SW = SK_ST / 64;      // threads / waves of the solve kernel
template <int PT>
__global__ void __launch_bounds__(SK_ST)
k_skel_solve(const SkelDev* __restrict__ dev, SkelClip* __restrict__ clip, double gtol, double* __restrict__ band_all,
             const double* __restrict__ rhs_all, double* __restrict__ yv_all, double* __restrict__ delta_all,
             const double* __restrict__ gn_part) {
  constexpr int LDP = PT + 1, NTP = PT / 16, RT = 4 * NTP;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // ---- the clip's iteration starts here: count it, and stop before any work when the projected gradient is small already
  //      (what the oracle's controller does at the top of an iteration)
  SkelClip& cs = clip[blockIdx.x];
  if (cs.status != 0) return;
  {
    __shared__ double gred[SK_ST / 64];
    __shared__ int stop;
    const int nfr = dev->n_frames;
    double gm = 0.0;
    for (int i = threadIdx.x; i < nfr; i += SK_ST) gm = fmax(gm, gn_part[(size_t)blockIdx.x * nfr + i]);
    for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
    if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SK_ST / 64; ++w) gm = fmax(gm, gred[w]);
      cs.gnorm = gm;
      cs.it += 1;
      cs.pivot_err = 0;
      stop = gm <= gtol;
      if (stop) cs.status = 3;
    }
    __syncthreads();
    if (stop) return;
  }
  int* const numeric_err = &cs.pivot_err;
  const size_t fr0 = (size_t)blockIdx.x * dev->n_frames;      // the clip's first frame
  double* const band = band_all + fr0 * 4 * PT * PT;
  const double* const rhs = rhs_all + fr0 * PT;
  double* const yv = yv_all + fr0 * PT;
  double* const delta = delta_all + fr0 * PT;
  double* Pn = reinterpret_cast<double*>(smem_raw);       // [4 PT][LDP]
  double* ring = Pn + 4 * PT * LDP;                        // [4][PT] right-hand sides / solutions of frames n .. n + 3
  double* tv = ring + 4 * PT;                              // [PT]
  const int N = dev->n_frames;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  auto load_panel = [&](int n) {
    for (int e = tid; e < 4 * PT * PT; e += SK_ST) {
      const int j = e / (PT * PT), rem = e % (PT * PT);
      Pn[(j * PT + rem / PT) * LDP + rem % PT] = (n + j < N) ? band[((size_t)n * 4 + j) * PT * PT + rem] : 0.0;
    }
  };
  for (int e = tid; e < 3 * PT; e += SK_ST) ring[e] = (e / PT < N) ? rhs[e] : 0.0;      // frames 0, 1, 2
  __syncthreads();
  for (int n = 0; n < N; ++n) {
    load_panel(n);
    if (tid < PT) ring[((n + 3) & 3) * PT + tid] = (n + 3 < N) ? rhs[(size_t)(n + 3) * PT + tid] : 0.0;
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < NTP; ++kb) {
      double* Tkk = Pn + (kb * 16) * LDP + kb * 16;
      if (wave == 0) {
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Tkk[(lk + 4 * r) * LDP + li];
        chol16_inv_acc<LDP>(Tkk, acc, lane, numeric_err);
      }
      __syncthreads();
      for (int t = kb + 1 + wave; t < RT; t += SK_SW) {          // panel: tile(t, kb) <- tile(t, kb) U_kk
        double* At = Pn + (t * 16) * LDP + kb * 16;
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          av[s] = At[li * LDP + 4 * s + lk];
          bv[s] = Tkk[(4 * s + lk) * LDP + li];
        }
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(av[s], bv[s], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) At[(lk + 4 * rr) * LDP + li] = acc[rr];
      }
      __syncthreads();
      int q = 0;                                             // trailing tiles inside the panel
      for (int ct = kb + 1; ct < NTP; ++ct)
        for (int rt = ct; rt < RT; ++rt, ++q) {
          if (q % SK_SW != wave) continue;
          double* Cc = Pn + (rt * 16) * LDP + ct * 16;
          const double* Ar = Pn + (rt * 16) * LDP + kb * 16;
          const double* Ac = Pn + (ct * 16) * LDP + kb * 16;
          d4 a;
          double av[4], bv[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(lk + 4 * rr) * LDP + li];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            av[s] = Ar[li * LDP + 4 * s + lk];
            bv[s] = Ac[li * LDP + 4 * s + lk];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) a = mfma(-av[s], bv[s], a);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LDP + li] = a[rr];
        }
      __syncthreads();
    }
    // ---- forward substitution with the block just factored
    double* rn = ring + (n & 3) * PT;
    for (int kb = 0; kb < NTP; ++kb) {
      if (tid >= 16 * kb && tid < 16 * kb + 16) {
        double u = rn[tid];
        for (int c = 0; c < 16 * kb; ++c) u -= Pn[tid * LDP + c] * rn[c];
        tv[tid] = u;
      }
      __syncthreads();
      if (tid >= 16 * kb && tid < 16 * kb + 16) {             // y = U_kk^T u  (U upper triangular)
        double y = 0.0;
        for (int r = 16 * kb; r <= tid; ++r) y += Pn[r * LDP + tid] * tv[r];
        rn[tid] = y;
      }
      __syncthreads();
    }
    if (tid < PT) yv[(size_t)n * PT + tid] = rn[tid];
    if (tid < 3 * PT) {
      const int j = 1 + tid / PT, i = tid % PT;
      if (n + j < N) {
        double s = 0.0;
        for (int c = 0; c < PT; ++c) s += Pn[(j * PT + i) * LDP + c] * rn[c];
        ring[((n + j) & 3) * PT + i] -= s;
      }
    }
    // ---- window update in memory: block (n + i, n + j) -= L_i L_j^T, stored at band[n + j][i - j].  Every wave requests ALL
    //      its tiles first (one round trip to memory instead of one per tile), then multiplies, then stores.
    {
      constexpr int WT = 6 * NTP * NTP, PER = (WT + SK_SW - 1) / SK_SW;
      d4 acc[PER];
      auto tile_of = [&](int t, int& i, int& j, int& rt, int& ct) {
        const int blk = t / (NTP * NTP), rem = t % (NTP * NTP);
        i = blk < 1 ? 1 : (blk < 3 ? 2 : 3);
        j = blk < 1 ? 1 : (blk < 3 ? blk : blk - 2);
        rt = rem / NTP;
        ct = rem % NTP;
      };
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int t = wave + SK_SW * q;
        int i = 1, j = 1, rt = 0, ct = 0;
        tile_of(t < WT ? t : 0, i, j, rt, ct);
        if (t < WT && n + i < N) {
          const double* Cg = band + ((size_t)(n + j) * 4 + (i - j)) * PT * PT;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) acc[q][rr] = Cg[(rt * 16 + lk + 4 * rr) * PT + ct * 16 + li];
        }
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int t = wave + SK_SW * q;
        int i = 1, j = 1, rt = 0, ct = 0;
        tile_of(t < WT ? t : 0, i, j, rt, ct);
        if (t < WT && n + i < N) {
          const double* Ar = Pn + (i * PT + rt * 16) * LDP;
          const double* Ac = Pn + (j * PT + ct * 16) * LDP;
          d4 a = acc[q];
#pragma unroll
          for (int s4 = 0; s4 < PT / 4; ++s4) a = mfma(-Ar[li * LDP + 4 * s4 + lk], Ac[li * LDP + 4 * s4 + lk], a);
          double* Cg = band + ((size_t)(n + j) * 4 + (i - j)) * PT * PT;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) Cg[(rt * 16 + lk + 4 * rr) * PT + ct * 16 + li] = a[rr];
        }
      }
    }
    // ---- the factored panel replaces the frame's blocks (read again by the backward pass)
    for (int e = tid; e < 4 * PT * PT; e += SK_ST) {
      const int j = e / (PT * PT), rem = e % (PT * PT);
      if (n + j < N || j == 0) band[((size_t)n * 4 + j) * PT * PT + rem] = Pn[(j * PT + rem / PT) * LDP + rem % PT];
    }
    __syncthreads();
  }
  // ---------------- backward ----------------
  for (int e = tid; e < 4 * PT; e += SK_ST) ring[e] = 0.0;
  __syncthreads();
  for (int n = N - 1; n >= 0; --n) {
    load_panel(n);
    __syncthreads();
    if (tid < PT) {
      double t = yv[(size_t)n * PT + tid];
      for (int j = 1; j < 4; ++j) {
        if (n + j >= N) break;
        const double* xj = ring + ((n + j) & 3) * PT;
        for (int r = 0; r < PT; ++r) t -= Pn[(j * PT + r) * LDP + tid] * xj[r];
      }
      tv[tid] = t;
    }
    __syncthreads();
    double* xn = ring + (n & 3) * PT;
    for (int kb = NTP - 1; kb >= 0; --kb) {
      if (tid >= 16 * kb && tid < 16 * kb + 16) {
        double s = tv[tid];
        for (int r = 16 * (kb + 1); r < PT; ++r) s -= Pn[r * LDP + tid] * xn[r];
        tv[tid] = s;
      }
      __syncthreads();
      if (tid >= 16 * kb && tid < 16 * kb + 16) {             // x = U_kk s
        double xx = 0.0;
        for (int c = tid; c < 16 * kb + 16; ++c) xx += Pn[tid * LDP + c] * tv[c];
        xn[tid] = xx;
      }
      __syncthreads();
    }
    if (tid < PT) delta[(size_t)n * PT + tid] = xn[tid];
    __syncthreads();
  }
}

// ---- the same solve with the window in REGISTERS (round 6) ---------------------------------------------------------------
// k_skel_solve walks the band through memory: every frame reloads its 4 PT x PT panel (73 KB at PT = 48), reads, updates and
// writes back the six window blocks, and writes the panel again - three dependent HBM round trips per frame - and its
// substitutions are loops of one thread per row.  Here the six window blocks (n + i, n + j), 1 <= j <= i <= 3, stay in the
// accumulator registers of the four waves (tile (rt, ct) of every block on wave (rt NTP + ct) % 4: the shift from one frame to
// the next then never leaves a wave), the next frame's panel is written from them into LDS, the four blocks of row n + 4 that
// enter the window are requested a frame ahead, and only the factored panel goes to memory (for the backward pass, whose panels
// are requested a frame ahead as well).  The substitutions: the 16 x 16 products with the inverted diagonal tiles on 256
// threads + a shuffle tree, the updates of the rows below one thread per row.  Same arithmetic per entry, other summation order.
constexpr int SK2_T = 256, SK2_W = SK2_T / 64;     // four waves: 512 registers per lane hold the window (eight waves spill it)
static_assert(SK2_T == 256, "the 16 x 16 products of the substitutions are laid out on 256 threads");
template <int PT>
__global__ void __launch_bounds__(SK2_T)
k_skel_solve2(const SkelDev* __restrict__ dev, SkelClip* __restrict__ clip, double gtol, double* __restrict__ band_all,
              const double* __restrict__ rhs_all, double* __restrict__ yv_all, double* __restrict__ delta_all,
              const double* __restrict__ gn_part) {
  constexpr int LDP = PT + 1, NTP = PT / 16, RT = 4 * NTP, NT2 = NTP * NTP, NS = (NT2 + SK2_W - 1) / SK2_W;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SkelClip& cs = clip[blockIdx.x];
  if (cs.status != 0) return;
  {
    __shared__ double gred[SK2_T / 64];
    __shared__ int stop;
    const int nfr = dev->n_frames;
    double gm = 0.0;
    for (int i = threadIdx.x; i < nfr; i += SK2_T) gm = fmax(gm, gn_part[(size_t)blockIdx.x * nfr + i]);
    for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
    if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SK2_T / 64; ++w) gm = fmax(gm, gred[w]);
      cs.gnorm = gm;
      cs.it += 1;
      cs.pivot_err = 0;
      stop = gm <= gtol;
      if (stop) cs.status = 3;
    }
    __syncthreads();
    if (stop) return;
  }
  int* const numeric_err = &cs.pivot_err;
  const size_t fr0 = (size_t)blockIdx.x * dev->n_frames;
  double* const band = band_all + fr0 * 4 * PT * PT;
  const double* const rhs = rhs_all + fr0 * PT;
  double* const yv = yv_all + fr0 * PT;
  double* const delta = delta_all + fr0 * PT;
  double* Pn = reinterpret_cast<double*>(smem_raw);       // [4 PT][LDP]
  double* ring = Pn + 4 * PT * LDP;                        // [4][PT] right-hand sides / solutions of frames n .. n + 3
  double* tv = ring + 4 * PT;                              // [PT]
  double* part = tv + PT;                                  // [10][PT] partial sums of the backward pass
  const int N = dev->n_frames;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  // block (row frame a, column frame b) of the band, b <= a <= b + 3 (nullptr beyond the clip: zeros)
  auto blockp = [&](int a, int b) -> const double* { return (a < N && b < N) ? band + ((size_t)b * 4 + (a - b)) * PT * PT : nullptr; };
  // Tiles to waves.  The six window blocks form three chains under the shift of a frame: A = (3,3) -> (2,2) -> (1,1) -> panel (the
  // DIAGONAL blocks: lower tiles only), B = (3,2) -> (2,1) -> panel, C = (3,1) -> panel.  Tile t of a chain's block lives on wave
  // (t + off) % 4, slot t / 4, with its own off per chain - the same for every block of the chain, so the shift never leaves a wave,
  // and the waves that get the odd tile differ: 11 / 12 / 12 / 10 tiles per wave at PT = 48 instead of 18 on wave 0.
  constexpr int NL = NTP * (NTP + 1) / 2;
  constexpr int OFF_A = 1, OFF_B = 0, OFF_C = 3;
  auto tile_rc = [&](int sl, int off, bool lower, int& rt, int& ct) {
    const int t = 4 * sl + ((wave - off) & 3);
    if (lower) {
      rt = (t >= 1) + (t >= 3) + (t >= 6);
      ct = t - rt * (rt + 1) / 2;
      return t < NL;
    }
    rt = t / NTP;
    ct = t % NTP;
    return t < NT2;
  };
  auto tile_load = [&](const double* B, int sl, int off, bool lower) {
    int rt, ct;
    d4 v = {0, 0, 0, 0};
    if (tile_rc(sl, off, lower, rt, ct) && B) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) v[rr] = B[(rt * 16 + lk + 4 * rr) * PT + ct * 16 + li];
    }
    return v;
  };
  auto tile_to_panel = [&](int row_block, int sl, int off, bool lower, const d4& v) {
    int rt, ct;
    if (tile_rc(sl, off, lower, rt, ct)) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Pn[(row_block * PT + rt * 16 + lk + 4 * rr) * LDP + ct * 16 + li] = v[rr];
    }
  };
  // window blocks in registers: 0 (1,1)  1 (2,1)  2 (2,2)  3 (3,1)  4 (3,2)  5 (3,3); fresh: row n + 4, columns n + 1 .. n + 4
  // (fresh[0] goes to the panel, [1] into chain C, [2] into chain B, [3] into chain A)
  constexpr int woff[6] = {OFF_A, OFF_B, OFF_A, OFF_C, OFF_B, OFF_A}, foff[4] = {OFF_B, OFF_C, OFF_B, OFF_A};
  constexpr bool wlow[6] = {true, false, true, false, false, true}, flow[4] = {false, false, false, true};
  d4 win[6][NS], fresh[4][NS];
  {
    const int wi[6] = {1, 2, 2, 3, 3, 3}, wj[6] = {1, 1, 2, 1, 2, 3};
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) win[b][sl] = tile_load(blockp(wi[b], wj[b]), sl, woff[b], wlow[b]);
  }
  for (int e = tid; e < 4 * PT * PT; e += SK2_T) {          // the panel of frame 0
    const int j = e / (PT * PT), rem = e % (PT * PT);
    Pn[(j * PT + rem / PT) * LDP + rem % PT] = (j < N) ? band[(size_t)j * PT * PT + rem] : 0.0;
  }
  for (int e = tid; e < 3 * PT; e += SK2_T) ring[e] = (e / PT < N) ? rhs[e] : 0.0;      // frames 0, 1, 2
  __syncthreads();
  for (int n = 0; n < N; ++n) {
    // the four blocks of row n + 4 (untouched so far: nothing reaches further than three frames), a frame ahead of their use
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) fresh[j][sl] = tile_load(blockp(n + 4, n + 1 + j), sl, foff[j], flow[j]);
    if (tid < PT) ring[((n + 3) & 3) * PT + tid] = (n + 3 < N) ? rhs[(size_t)(n + 3) * PT + tid] : 0.0;
    // ---- factor the panel (as k_skel_solve, with a LOOK-AHEAD: while waves 1 .. 3 do the trailing tiles of block column kb, wave
    //      0 updates the next diagonal tile alone and goes straight into its 16-pivot chain - the chains but the first are off the
    //      other waves' critical path)
    if (wave == 0) {
      double* T00 = Pn;
      d4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = T00[(lk + 4 * r) * LDP + li];
      chol16_inv_acc<LDP>(T00, acc, lane, numeric_err);
    }
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < NTP; ++kb) {
      double* Tkk = Pn + (kb * 16) * LDP + kb * 16;
      for (int t = kb + 1 + wave; t < RT; t += SK2_W) {          // panel: tile(t, kb) <- tile(t, kb) U_kk
        double* At = Pn + (t * 16) * LDP + kb * 16;
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          av[s] = At[li * LDP + 4 * s + lk];
          bv[s] = Tkk[(4 * s + lk) * LDP + li];
        }
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(av[s], bv[s], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) At[(lk + 4 * rr) * LDP + li] = acc[rr];
      }
      __syncthreads();
      if (kb + 1 == NTP) break;
      if (wave == 0) {                                       // the next diagonal tile and its chain
        double* Cc = Pn + ((kb + 1) * 16) * LDP + (kb + 1) * 16;
        const double* A = Pn + ((kb + 1) * 16) * LDP + kb * 16;
        d4 a;
        double av[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(lk + 4 * rr) * LDP + li];
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = A[li * LDP + 4 * s + lk];
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma(-av[s], av[s], a);
        chol16_inv_acc<LDP>(Cc, a, lane, numeric_err);
      } else {
        int q = 0;                                           // the other trailing tiles inside the panel: waves 1 .. 3
        for (int ct = kb + 1; ct < NTP; ++ct)
          for (int rt = ct; rt < RT; ++rt) {
            if (rt == kb + 1 && ct == kb + 1) continue;      // (wave 0's)
            if ((q++) % (SK2_W - 1) != wave - 1) continue;
            double* Cc = Pn + (rt * 16) * LDP + ct * 16;
            const double* Ar = Pn + (rt * 16) * LDP + kb * 16;
            const double* Ac = Pn + (ct * 16) * LDP + kb * 16;
            d4 a;
            double av[4], bv[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(lk + 4 * rr) * LDP + li];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              av[s] = Ar[li * LDP + 4 * s + lk];
              bv[s] = Ac[li * LDP + 4 * s + lk];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) a = mfma(-av[s], bv[s], a);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LDP + li] = a[rr];
          }
      }
      __syncthreads();
    }
    // ---- the factored panel to memory (the backward pass reads it): requested now, drains under what follows
    for (int e = tid; e < 4 * PT * PT; e += SK2_T) {
      const int j = e / (PT * PT), rem = e % (PT * PT);
      if (n + j < N) band[((size_t)n * 4 + j) * PT * PT + rem] = Pn[(j * PT + rem / PT) * LDP + rem % PT];
    }
    // ---- forward substitution, block by block: y_kb = U_kk^T u_kb (a 16 x 16 product: threads (r, c) + a shuffle tree over c),
    //      then every row below loses L[row][kb block] y_kb - the rows of this frame and the three frames' right-hand sides
    double* rn = ring + (n & 3) * PT;
    for (int kb = 0; kb < NTP; ++kb) {
      {
        const int r = tid >> 4, c = tid & 15;                // y[r] = sum_c U[c][r] u[c]   (U upper triangular: c <= r)
        double v = c <= r ? Pn[(16 * kb + c) * LDP + 16 * kb + r] * rn[16 * kb + c] : 0.0;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (c == 0) tv[16 * kb + r] = v;
      }
      __syncthreads();
      if (tid < 16) rn[16 * kb + tid] = tv[16 * kb + tid];
      if (tid >= 16 * (kb + 1) && tid < 4 * PT) {              // row tid of the panel (rows >= PT: frames n + 1 .. n + 3)
        const int row = tid;
        double sub = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) sub += Pn[row * LDP + 16 * kb + c] * tv[16 * kb + c];
        if (row < PT) rn[row] -= sub;
        else ring[((n + row / PT) & 3) * PT + row % PT] -= sub;
      }
      __syncthreads();
    }
    if (tid < PT) yv[(size_t)n * PT + tid] = rn[tid];
    // ---- window update in registers: block (n + i, n + j) -= L_i L_j^T
    {
      const int wi[6] = {1, 2, 2, 3, 3, 3}, wj[6] = {1, 1, 2, 1, 2, 3};
#pragma unroll
      for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          int rt, ct;
          if (tile_rc(sl, woff[b], wlow[b], rt, ct)) {
            const double* Ar = Pn + (wi[b] * PT + rt * 16 + li) * LDP + lk;
            const double* Ac = Pn + (wj[b] * PT + ct * 16 + li) * LDP + lk;
            win[b][sl] = mma_seq<PT / 4, true>(win[b][sl], Ar, 4, Ac, 4);
          }
        }
    }
    __syncthreads();                                         // every read of this frame's panel is done
    // ---- the next frame's panel from the window; the window moves on by one frame
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      tile_to_panel(0, sl, OFF_A, true, win[0][sl]);         // (the diagonal block: lower tiles - nothing reads the others)
      tile_to_panel(1, sl, OFF_B, false, win[1][sl]);
      tile_to_panel(2, sl, OFF_C, false, win[3][sl]);
      tile_to_panel(3, sl, OFF_B, false, fresh[0][sl]);
      win[0][sl] = win[2][sl];
      win[1][sl] = win[4][sl];
      win[2][sl] = win[5][sl];
      win[3][sl] = fresh[1][sl];
      win[4][sl] = fresh[2][sl];
      win[5][sl] = fresh[3][sl];
    }
    __syncthreads();
  }
  // ---------------- backward: x_n = L_nn^-T (y_n - sum_j L_n+j,n^T x_n+j) ----------------
  constexpr int PV = (4 * PT * PT + SK2_T - 1) / SK2_T;
  double pre[PV];
  auto request_panel = [&](int n) {
#pragma unroll
    for (int k = 0; k < PV; ++k) {
      const int e = tid + SK2_T * k;
      const int j = e / (PT * PT);
      pre[k] = (e < 4 * PT * PT && n >= 0 && n + j < N) ? band[(size_t)n * 4 * PT * PT + e] : 0.0;
    }
  };
  auto stage_panel = [&]() {
#pragma unroll
    for (int k = 0; k < PV; ++k) {
      const int e = tid + SK2_T * k;
      if (e < 4 * PT * PT) {
        const int j = e / (PT * PT), rem = e % (PT * PT);
        Pn[(j * PT + rem / PT) * LDP + rem % PT] = pre[k];
      }
    }
  };
  for (int e = tid; e < 4 * PT; e += SK2_T) ring[e] = 0.0;
  request_panel(N - 1);
  __syncthreads();
  for (int n = N - 1; n >= 0; --n) {
    stage_panel();
    request_panel(n - 1);                                    // a frame ahead
    const double yn = tid < PT ? yv[(size_t)n * PT + tid] : 0.0;
    __syncthreads();
    // t = y_n - sum_j L_j^T x_n+j: 9 partial sums per column (3 frames x 3 thirds of the rows), then one add per column
    for (int t9 = tid; t9 < 9 * PT; t9 += SK2_T) {
      const int col = t9 % PT, pj = t9 / PT, j = 1 + pj / 3, third = pj % 3;
      constexpr int R3 = (PT + 2) / 3;
      const double* xj = ring + ((n + j) & 3) * PT;
      double sm = 0.0;
      if (n + j < N)
        for (int r = third * R3; r < min(PT, (third + 1) * R3); ++r) sm += Pn[(j * PT + r) * LDP + col] * xj[r];
      part[pj * PT + col] = sm;
    }
    __syncthreads();
    if (tid < PT) {
      double t = yn;
#pragma unroll
      for (int q = 0; q < 9; ++q) t -= part[q * PT + tid];
      tv[tid] = t;
    }
    __syncthreads();
    double* xn = ring + (n & 3) * PT;
    for (int kb = NTP - 1; kb >= 0; --kb) {
      {                                                       // x[r] = sum_c U[r][c] s[c]   (c >= r)
        const int r = tid >> 4, c = tid & 15;
        double v = c >= r ? Pn[(16 * kb + r) * LDP + 16 * kb + c] * tv[16 * kb + c] : 0.0;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (c == 0) xn[16 * kb + r] = v;
      }
      __syncthreads();
      if (tid < 16 * kb) {                                    // the columns above lose L[kb block][col]^T x_kb
        double sub = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) sub += Pn[(16 * kb + r) * LDP + tid] * xn[16 * kb + r];
        tv[tid] -= sub;
      }
      __syncthreads();
    }
    if (tid < PT) delta[(size_t)n * PT + tid] = xn[tid];
    __syncthreads();
  }
}

// ---- trial iterate and the controller's sums -------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_skel_trial(const SkelDev* __restrict__ dev, const SkelClip* __restrict__ clip, double* __restrict__ xb0,
             double* __restrict__ xb1, const double* __restrict__ g0, const double* __restrict__ g1,
             const double* __restrict__ hd0, const double* __restrict__ hd1, const double* __restrict__ lo_all,
             const double* __restrict__ hi_all, const double* __restrict__ delta_all, double* __restrict__ pred_part,
             double* __restrict__ step_part) {
  // grid (blocks per clip, clips): the partial sums of clip b are pred_part[b * gridDim.x ..]
  const SkelDev& D = *dev;
  const int tid = threadIdx.x, b = blockIdx.y;
  const SkelClip& cs = clip[b];
  if (cs.status != 0) return;
  const size_t off = (size_t)b * D.n_frames * D.n_act;
  const double* __restrict__ x = (cs.cur ? xb1 : xb0) + off;
  double* __restrict__ xt = (cs.cur ? xb0 : xb1) + off;
  const double* __restrict__ g = (cs.cur ? g1 : g0) + off;
  const double* __restrict__ hd = (cs.cur ? hd1 : hd0) + off;
  const double* __restrict__ lo = lo_all + off;
  const double* __restrict__ hi = hi_all + off;
  const double* __restrict__ delta = delta_all + (size_t)b * D.n_frames * D.PT;
  const double lam = cs.lam;
  const int64_t e = (int64_t)blockIdx.x * 256 + tid;
  double pred = 0.0, step = 0.0;
  if (e < (int64_t)D.n_frames * D.n_act) {
    const int n = (int)(e / D.n_act), p = (int)(e % D.n_act);
    const double xv = x[e], gv = g[e], d0 = hd[e];
    const bool fixed = skel_fixed(xv, gv, d0, lo[e], hi[e]);
    const double d = fixed ? 0.0 : delta[(size_t)n * D.PT + p], pg = fixed ? 0.0 : gv;
    const double xn = fmin(fmax(xv + d, lo[e]), hi[e]);
    xt[e] = xn;
    pred = 0.5 * d * (lam * fmax(d0, D.lam_floor) * d - pg);
    step = fabs(xn - xv);
  }
  __shared__ double rp[4], rs[4];
  for (int off = 32; off > 0; off >>= 1) {
    pred += __shfl_down(pred, off, 64);
    step = fmax(step, __shfl_down(step, off, 64));
  }
  if ((tid & 63) == 0) {
    rp[tid >> 6] = pred;
    rs[tid >> 6] = step;
  }
  __syncthreads();
  if (tid == 0) {
    pred_part[(size_t)b * gridDim.x + blockIdx.x] = (rp[0] + rp[1]) + (rp[2] + rp[3]);
    step_part[(size_t)b * gridDim.x + blockIdx.x] = fmax(fmax(rs[0], rs[1]), fmax(rs[2], rs[3]));
  }
}

// The controller of one clip (lm_control_local of fte_api.hip / oracle.fte.lm_solve): sums in a fixed order, then thread 0
// decides.  mode 0: the cost of the initial iterate; 1: an iteration's trial (cost of the trial buffer, predicted reduction,
// step length); 2: only the gradient norm of the final iterate (gn_part from a final k_skel_build).
// A failed factorisation (non-positive pivot at this damping) is a rejected step: lambda goes up, as in the cheetah path; only
// when lambda runs out is it reported as a numeric failure (status 5).
__global__ void __launch_bounds__(256)
k_skel_control(const SkelDev* __restrict__ dev, SkelClip* __restrict__ clip, int mode, const double* __restrict__ c0,
               const double* __restrict__ c1, const double* __restrict__ pred_part, const double* __restrict__ step_part,
               int n_trial, const double* __restrict__ gn_part, double lam0, double ftol, double xtol, double lam_max) {
  __shared__ double sh[4][4];
  const int b = blockIdx.x, N = dev->n_frames;
  SkelClip& cs = clip[b];
  if (mode == 1 && cs.status != 0) return;
  const double* cost_part = ((mode == 1 ? cs.cur ^ 1 : cs.cur) ? c1 : c0) + (size_t)b * N;
  double c = 0.0, p = 0.0, s = 0.0, g = 0.0;
  if (mode != 2)
    for (int i = threadIdx.x; i < N; i += 256) c += cost_part[i];
  if (mode == 1)
    for (int i = threadIdx.x; i < n_trial; i += 256) {
      p += pred_part[(size_t)b * n_trial + i];
      s = fmax(s, step_part[(size_t)b * n_trial + i]);
    }
  if (mode == 2)
    for (int i = threadIdx.x; i < N; i += 256) g = fmax(g, gn_part[(size_t)b * N + i]);
  for (int off = 32; off > 0; off >>= 1) {
    c += __shfl_down(c, off, 64);
    p += __shfl_down(p, off, 64);
    s = fmax(s, __shfl_down(s, off, 64));
    g = fmax(g, __shfl_down(g, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    double* w = sh[threadIdx.x >> 6];
    w[0] = c;
    w[1] = p;
    w[2] = s;
    w[3] = g;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  c = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
  p = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
  s = fmax(fmax(sh[0][2], sh[1][2]), fmax(sh[2][2], sh[3][2]));
  g = fmax(fmax(sh[0][3], sh[1][3]), fmax(sh[2][3], sh[3][3]));
  if (mode == 0) {
    cs.F = cs.cost0 = c;
    cs.lam = lam0;
    cs.nu = 2.0;
    return;
  }
  if (mode == 2) {
    cs.gnorm = g;
    return;
  }
  cs.Ft = c;
  cs.pred = p;
  cs.step = s;
  const double F = cs.F, Ft = c;
  if (cs.pivot_err == 0 && Ft < F) {
    const double gain = p > 0.0 ? (F - Ft) / p : -1.0, dF = F - Ft, t = 2.0 * gain - 1.0;
    cs.cur ^= 1;
    cs.F = Ft;
    cs.accepted += 1;
    cs.lam = cs.lam * fmax(1.0 / 3.0, 1.0 - t * t * t);
    cs.nu = 2.0;
    if (dF <= ftol * fabs(Ft)) cs.status = 1;
    else if (s <= xtol) cs.status = 2;
  } else {
    cs.lam *= cs.nu;
    cs.nu *= 2.0;
    if (cs.lam > lam_max) cs.status = cs.pivot_err ? 5 : 4;
  }
}

__global__ void k_skel_clip(const double* __restrict__ src, const double* __restrict__ lo, const double* __restrict__ hi,
                            double* __restrict__ dst, int64_t n_total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n_total) dst[e] = fmin(fmax(src[e], lo[e]), hi[e]);
}
// the clips' final iterates (each from its own current buffer) -> out[clips][N][P]
__global__ void k_skel_gather(const SkelDev* __restrict__ dev, const SkelClip* __restrict__ clip, const double* __restrict__ x0,
                              const double* __restrict__ x1, double* __restrict__ out, int64_t n_total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n_total) out[e] = (clip[e / ((int64_t)dev->n_frames * dev->n_act)].cur ? x1 : x0)[e];
}

// poses of active-state rows: pos[N][n_pose][3]
__global__ void __launch_bounds__(256)
k_skel_poses(const SkelDev* __restrict__ dev, const double* __restrict__ x, double* __restrict__ pos, int64_t n_total) {
  const SkelDev& D = *dev;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_total) return;
  const double* xs = x + n * D.n_act;
  double* out = pos + n * D.n_pose * 3;
  for (int s = 0; s < D.n_pose; ++s)
    for (int j = 0; j < 3; ++j) out[s * 3 + j] = xs[j];
  for (int k = 0; k < D.n_ops; ++k) {
    const acino_skel_op& o = D.op[k];
    double Rm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (o.flags & 2) {
      double s, c;
      sincos(D.amap[k][1] >= 0 ? xs[D.amap[k][1]] : 0.0, &s, &c);
      Rm[0][0] = c; Rm[0][2] = -s; Rm[2][0] = s; Rm[2][2] = c;
    }
    if (o.flags & 1) {
      double s, c;
      sincos(D.amap[k][0] >= 0 ? xs[D.amap[k][0]] : 0.0, &s, &c);
      for (int j = 0; j < 3; ++j) {
        const double r1 = Rm[1][j], r2 = Rm[2][j];
        Rm[1][j] = c * r1 + s * r2;
        Rm[2][j] = -s * r1 + c * r2;
      }
    }
    if (o.flags & 4) {
      double s, c;
      sincos(D.amap[k][2] >= 0 ? xs[D.amap[k][2]] : 0.0, &s, &c);
      for (int j = 0; j < 3; ++j) {
        const double r0 = Rm[0][j], r1 = Rm[1][j];
        Rm[0][j] = c * r0 + s * r1;
        Rm[1][j] = -s * r0 + c * r1;
      }
    }
    for (int i = 0; i < 3; ++i) {
      const double d = (o.flags & 8) ? Rm[i][0] * o.off[0] + Rm[i][1] * o.off[1] + Rm[i][2] * o.off[2]
                                     : Rm[0][i] * o.off[0] + Rm[1][i] * o.off[1] + Rm[2][i] * o.off[2];
      out[o.child * 3 + i] = out[o.parent * 3 + i] + d;
    }
  }
}

struct SkelLayout {
  size_t dev, clip, x[2], g[2], H[2], hd[2], cost[2], band, rhs, yv, delta, pred, step, gn, total;
};
static size_t sk_align(size_t v) { return (v + 255) / 256 * 256; }
// (N: frames of ALL clips)
static SkelLayout skel_layout(size_t N, int n_clips, int frames_per_clip, int P, int PT) {
  SkelLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = sk_align(off + bytes);
    return o;
  };
  L.dev = take(sizeof(SkelDev));
  L.clip = take(sizeof(SkelClip) * (size_t)n_clips);
  for (int k = 0; k < 2; ++k) L.x[k] = take(sizeof(double) * N * P);
  for (int k = 0; k < 2; ++k) L.g[k] = take(sizeof(double) * N * P);
  for (int k = 0; k < 2; ++k) L.H[k] = take(sizeof(double) * N * P * P);
  for (int k = 0; k < 2; ++k) L.hd[k] = take(sizeof(double) * N * P);
  for (int k = 0; k < 2; ++k) L.cost[k] = take(sizeof(double) * N);
  L.band = take(sizeof(double) * N * 4 * PT * PT);
  L.rhs = take(sizeof(double) * (N + 4) * PT);
  L.yv = take(sizeof(double) * N * PT);
  L.delta = take(sizeof(double) * N * PT);
  const size_t nt = (((size_t)frames_per_clip * P + 255) / 256) * (size_t)n_clips + 1;
  L.pred = take(sizeof(double) * nt);
  L.step = take(sizeof(double) * nt);
  L.gn = take(sizeof(double) * N);
  L.total = off;
  return L;
}

static int skel_validate(const acino_skel_fte_params* p) {
  ACINO_REQUIRE(p != nullptr, "params");
  ACINO_REQUIRE(p->n_frames >= 1, "n_frames >= 1");
  ACINO_REQUIRE(p->n_cams >= 1 && p->n_cams <= ACINO_MAX_CAMS, "n_cams in 1..16");
  ACINO_REQUIRE(p->n_pose >= 1 && p->n_pose <= ACINO_SKEL_MAX_OPS + 1, "n_pose");
  ACINO_REQUIRE(p->n_ops >= 0 && p->n_ops <= ACINO_SKEL_MAX_OPS, "n_ops <= ACINO_SKEL_MAX_OPS");
  ACINO_REQUIRE(p->n_angles >= 1, "n_angles");
  ACINO_REQUIRE(p->n_active >= 3 && p->n_active <= SK_MAXP, "n_active in 3..64 (x, y, z and the angles that move a pose)");
  ACINO_REQUIRE(2 * p->n_pose * p->n_cams <= SK_MAXROWS, "2 * n_pose * n_cams <= 256 residual rows per frame");
  ACINO_REQUIRE(p->n_pose * p->n_cams <= 256, "n_pose * n_cams <= 256");
  ACINO_REQUIRE(p->h > 0 && p->model_weight >= 0 && p->l1_eps > 0, "h > 0, model_weight >= 0, l1_eps > 0");
  ACINO_REQUIRE(p->lam0 > 0 && p->max_iter >= 0, "lam0 > 0, max_iter >= 0");
  return ACINO_OK;
}

}  // namespace acino

using namespace acino;

extern "C" {

size_t acino_sizeof_skel_fte_params(void) { return sizeof(acino_skel_fte_params); }
size_t acino_sizeof_skel_fte_info(void) { return sizeof(acino_skel_fte_info); }

size_t acino_skel_fte_workspace_bytes_batch(const acino_skel_fte_params* p, int n_clips) {
  if (!p || p->n_frames < 1 || p->n_active < 3 || p->n_active > SK_MAXP || n_clips < 1) return 0;
  const int PT = (p->n_active + 15) / 16 * 16;
  return skel_layout((size_t)p->n_frames * n_clips, n_clips, p->n_frames, p->n_active, PT).total + 256;
}
size_t acino_skel_fte_workspace_bytes(const acino_skel_fte_params* p) { return acino_skel_fte_workspace_bytes_batch(p, 1); }

int acino_skel_fte_solve_batch(const acino_skel_fte_params* p, int n_clips, const acino_skel_op* h_ops, const int32_t* h_active,
                               const double* d_meas, const double* d_w, const double* d_cams24, const double* d_lo,
                               const double* d_hi, double* d_x, double* d_pos, void* d_workspace, size_t workspace_bytes,
                               acino_skel_fte_info* infos, void* stream) {
  int rc = skel_validate(p);
  if (rc) return rc;
  ACINO_REQUIRE(n_clips >= 1 && n_clips <= 65535, "n_clips in 1..65535");
  ACINO_REQUIRE(h_ops && h_active && d_meas && d_w && d_cams24 && d_lo && d_hi && d_x && d_workspace, "null buffer");
  ACINO_REQUIRE(((uintptr_t)d_workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int N = p->n_frames, B = n_clips, P = p->n_active, PT = (P + 15) / 16 * 16, L = p->n_angles;
  const size_t NT = (size_t)N * B;                           // frames of all clips
  ACINO_REQUIRE(NT < (size_t)1 << 31, "n_clips * n_frames < 2^31");
  const SkelLayout lay = skel_layout(NT, B, N, P, PT);
  ACINO_REQUIRE(workspace_bytes >= lay.total, "workspace too small (acino_skel_fte_workspace_bytes_batch)");
  // ---- the program: active index of every op's parent angles, the ops on every pose's path
  std::vector<SkelDev> hv(1);
  SkelDev& h = hv[0];
  memset(&h, 0, sizeof(h));
  h.n_frames = N;
  h.n_cams = p->n_cams;
  h.n_pose = p->n_pose;
  h.n_ops = p->n_ops;
  h.n_act = P;
  h.PT = PT;
  h.n_rows = 2 * p->n_pose * p->n_cams;
  h.q = p->model_weight / (p->h * p->h * p->h * p->h);
  h.l1_eps = p->l1_eps;
  h.lam_floor = DIAG_FLOOR;
  ACINO_REQUIRE(h_active[0] == 0 && h_active[1] == 1 && h_active[2] == 2, "the first three active states are x, y, z");
  std::vector<int> where(3 + 3 * L, -1);
  for (int a = 0; a < P; ++a) {
    ACINO_REQUIRE(h_active[a] >= 0 && h_active[a] < 3 + 3 * L && (a == 0 || h_active[a] > h_active[a - 1]),
                  "active state indices must be increasing and inside [0, 3 + 3 L)");
    where[h_active[a]] = a;
  }
  std::vector<unsigned long long> path(p->n_pose, 0ull);
  for (int k = 0; k < p->n_ops; ++k) {
    const acino_skel_op& o = h_ops[k];
    ACINO_REQUIRE(o.child >= 0 && o.child < p->n_pose && o.parent >= 0 && o.parent < p->n_pose, "op slot out of range");
    ACINO_REQUIRE(o.angle >= 0 && o.angle < L, "op angle index out of range");
    h.op[k] = o;
    for (int ax = 0; ax < 3; ++ax) {
      const int st = 3 + ax * L + o.angle;
      const int a = ((o.flags >> ax) & 1) ? where[st] : -1;
      ACINO_REQUIRE(!((o.flags >> ax) & 1) || a >= 0, "an enabled angle of a parent part is missing from the active states");
      h.amap[k][ax] = (int8_t)a;
    }
    h.amap[k][3] = -1;
    path[o.child] = path[o.parent] | (1ull << k);          // (a slot defined twice keeps its last definition, as the poses)
  }
  for (int l = 0; l < p->n_pose; ++l) h.pmask[l] = path[l];
  char* base = (char*)d_workspace;
  auto D = [&](size_t off) { return reinterpret_cast<double*>(base + off); };
  SkelDev* d_dev = reinterpret_cast<SkelDev*>(base + lay.dev);
  SkelClip* d_clip = reinterpret_cast<SkelClip*>(base + lay.clip);
  ACINO_HIP_CHECK(hipMemcpyAsync(d_dev, &h, sizeof(SkelDev), hipMemcpyHostToDevice, s));
  ACINO_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char*>(d_dev) + offsetof(SkelDev, cams), d_cams24,
                                 sizeof(double) * ACINO_CAM_STRIDE * p->n_cams, hipMemcpyDeviceToDevice, s));
  ACINO_HIP_CHECK(hipMemsetAsync(d_clip, 0, sizeof(SkelClip) * (size_t)B, s));
  ACINO_HIP_CHECK(hipStreamSynchronize(s));                // (h lives on this frame)
  const size_t lds_asm = sizeof(double) * (SK_MAXP + ACINO_SKEL_MAX_OPS * 12 + (ACINO_SKEL_MAX_OPS + 1) * 3 + SK_MAXROWS * 5 + 8 +
                                           (size_t)h.n_rows * (P | 1));
  const size_t lds_solve = sizeof(double) * ((size_t)4 * PT * (PT + 1) + 16 * PT);     // panel, ring, tv, the backward pass's partial sums
  static const bool solve_in_registers = getenv("ACINO_SKEL_OLD_SOLVE") == nullptr;     // (A/B switch: the round-5 kernel walks the band through memory)
  {
    static PerDeviceOnce attr;
    if (attr.first()) {
      const int big = 160 * 1024, big_solve = 160 * 1024 - 1024;   // (the solve kernel also has a few static words)
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_assemble<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_assemble<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve<16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve<32>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve<48>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve<64>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve2<16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve2<32>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve2<48>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
      ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_skel_solve2<64>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, big_solve));
    }
  }
  ACINO_REQUIRE(lds_asm <= 160 * 1024, "residual rows x active states do not fit the assembly's LDS");
  const int n_trial = (int)(((size_t)N * P + 255) / 256);    // blocks per clip
  const int64_t n_el = (int64_t)NT * P;
  const double lam_max = p->lam_max > 0 ? p->lam_max : 1e16;
  auto assemble = [&](int which) -> int {
    hipLaunchKernelGGL(k_skel_assemble<true>, dim3((unsigned)NT), dim3(256), lds_asm, s, d_dev, d_clip, which, D(lay.x[0]),
                       D(lay.x[1]), d_meas, d_w, D(lay.H[0]), D(lay.H[1]), D(lay.g[0]), D(lay.g[1]), D(lay.hd[0]), D(lay.hd[1]),
                       D(lay.cost[0]), D(lay.cost[1]));
    ACINO_LAUNCH_CHECK();
    return ACINO_OK;
  };
  auto build = [&](int final_pass) -> int {
    hipLaunchKernelGGL(k_skel_build, dim3((unsigned)NT), dim3(256), 0, s, d_dev, d_clip, D(lay.x[0]), D(lay.x[1]), D(lay.g[0]),
                       D(lay.g[1]), D(lay.H[0]), D(lay.H[1]), D(lay.hd[0]), D(lay.hd[1]), d_lo, d_hi, D(lay.band), D(lay.rhs),
                       D(lay.gn), final_pass);
    ACINO_LAUNCH_CHECK();
    return ACINO_OK;
  };
  auto control = [&](int mode) -> int {
    hipLaunchKernelGGL(k_skel_control, dim3(B), dim3(256), 0, s, d_dev, d_clip, mode, D(lay.cost[0]), D(lay.cost[1]), D(lay.pred),
                       D(lay.step), n_trial, D(lay.gn), p->lam0, p->ftol, p->xtol, lam_max);
    ACINO_LAUNCH_CHECK();
    return ACINO_OK;
  };
  hipLaunchKernelGGL(k_skel_clip, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, d_x, d_lo, d_hi, D(lay.x[0]), n_el);
  ACINO_LAUNCH_CHECK();
  if ((rc = assemble(0))) return rc;
  if ((rc = control(0))) return rc;
  // ---- the iterations: every clip's controller is on the device; the host reads the status words to know when to stop
  std::vector<SkelClip> hc(B);
  for (int it = 1; it <= p->max_iter; ++it) {
    if ((rc = build(0))) return rc;
    if (solve_in_registers && PT <= 48) {        // (PT = 64: window + operands exceed the 512 registers of a lane; the round-5 kernel)
      switch (PT) {
        case 16: hipLaunchKernelGGL(k_skel_solve2<16>, dim3(B), dim3(SK2_T), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
        case 32: hipLaunchKernelGGL(k_skel_solve2<32>, dim3(B), dim3(SK2_T), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
        case 48: hipLaunchKernelGGL(k_skel_solve2<48>, dim3(B), dim3(SK2_T), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
        default: hipLaunchKernelGGL(k_skel_solve2<64>, dim3(B), dim3(SK2_T), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
      }
    } else {
      switch (PT) {
        case 16: hipLaunchKernelGGL(k_skel_solve<16>, dim3(B), dim3(SK_ST), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
        case 32: hipLaunchKernelGGL(k_skel_solve<32>, dim3(B), dim3(SK_ST), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
        case 48: hipLaunchKernelGGL(k_skel_solve<48>, dim3(B), dim3(SK_ST), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
        default: hipLaunchKernelGGL(k_skel_solve<64>, dim3(B), dim3(SK_ST), lds_solve, s, d_dev, d_clip, p->gtol, D(lay.band), D(lay.rhs), D(lay.yv), D(lay.delta), D(lay.gn)); break;
      }
    }
    ACINO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_skel_trial, dim3(n_trial, B), dim3(256), 0, s, d_dev, d_clip, D(lay.x[0]), D(lay.x[1]), D(lay.g[0]),
                       D(lay.g[1]), D(lay.hd[0]), D(lay.hd[1]), d_lo, d_hi, D(lay.delta), D(lay.pred), D(lay.step));
    ACINO_LAUNCH_CHECK();
    if ((rc = assemble(1))) return rc;
    if ((rc = control(1))) return rc;
    ACINO_HIP_CHECK(hipMemcpyAsync(hc.data(), d_clip, sizeof(SkelClip) * (size_t)B, hipMemcpyDeviceToHost, s));
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    bool running = false;
    for (int b = 0; b < B; ++b) running = running || hc[b].status == 0;
    if (!running) break;
  }
  // ---- the gradient norm of the final iterates (a clip that stopped on ftol / xtol / max_iter holds the norm of the iterate
  //      BEFORE its last step; at max_iter = 0 none was ever computed), the results
  if ((rc = build(1))) return rc;
  if ((rc = control(2))) return rc;
  hipLaunchKernelGGL(k_skel_gather, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, d_dev, d_clip, D(lay.x[0]), D(lay.x[1]),
                     d_x, n_el);
  ACINO_LAUNCH_CHECK();
  if (d_pos) {
    hipLaunchKernelGGL(k_skel_poses, dim3((unsigned)((NT + 255) / 256)), dim3(256), 0, s, d_dev, d_x, d_pos, (int64_t)NT);
    ACINO_LAUNCH_CHECK();
  }
  ACINO_HIP_CHECK(hipMemcpyAsync(hc.data(), d_clip, sizeof(SkelClip) * (size_t)B, hipMemcpyDeviceToHost, s));
  ACINO_HIP_CHECK(hipStreamSynchronize(s));
  bool numeric = false;
  for (int b = 0; b < B; ++b) {
    numeric = numeric || hc[b].status == 5;
    if (infos) {
      infos[b].cost_initial = hc[b].cost0;
      infos[b].cost_final = hc[b].F;
      infos[b].gnorm_inf = hc[b].gnorm;
      infos[b].lam = hc[b].lam;
      infos[b].iterations = hc[b].it;
      infos[b].accepted = hc[b].accepted;
      infos[b].status = hc[b].status;
      infos[b].pad0 = 0;
    }
  }
  // One clip: the failure is the call's.  A batch: every clip's outcome is in infos[b].status (5 = numeric) and the results of
  // the other clips stand - one degenerate window must not cost the caller the whole video; without infos there is nowhere to
  // report it per clip, so the call fails as before.
  if (numeric && (B == 1 || !infos)) {
    set_error("non-positive pivot in the banded factorisation at every damping up to lam_max (system not positive definite)");
    return ACINO_ERR_NUMERIC;
  }
  return ACINO_OK;
}

int acino_skel_fte_solve(const acino_skel_fte_params* p, const acino_skel_op* h_ops, const int32_t* h_active,
                         const double* d_meas, const double* d_w, const double* d_cams24, const double* d_lo,
                         const double* d_hi, double* d_x, double* d_pos, void* d_workspace, size_t workspace_bytes,
                         acino_skel_fte_info* info, void* stream) {
  return acino_skel_fte_solve_batch(p, 1, h_ops, h_active, d_meas, d_w, d_cams24, d_lo, d_hi, d_x, d_pos, d_workspace,
                                    workspace_bytes, info, stream);
}

}  // extern "C"
