// Extended Kalman filter + Rauch-Tung-Striebel smoother over the 25 cheetah pose parameters (gfx950, fp64).
//
// Reference: `ekf` in src/all_optimizations.py:569-865 - constant-acceleration model (75 states), measurement =
// fisheye projection of the 20 markers in every camera (up to 240 pixel coordinates per frame), Jacobian by forward
// differences with eps = 1e-3 (:631-646), 3-sigma gating (:815-819), K = P H^T inv(S) with S 240x240 (:822),
// P <- (I - K H) P (:829), smoother (:836-841).  The filter is sequential in frames, so ONE workgroup runs a whole
// sequence (grid = sequences) with the 75x75 covariance, the 240x25 Jacobian and every intermediate in LDS:
//   * FK of the 26 forward-difference variants column-parallel (78 threads); projections only of the (variant,
//     marker) pairs in which the marker moves with the perturbed parameter (about half; found once per launch by
//     probing the chain at a generic pose - the other forward differences are exactly 0 at every pose);
//   * H has 25 non-zero columns and R is diagonal, so with M = Hq^T R^-1 Hq, g = Hq^T R^-1 r (fp64 MFMA, g rides
//     along as a 26th column) the 240x240 inverse collapses to 25x25 algebra:
//       K r = P[:, :25] (I + M Pq)^-1 g,   P <- P - P[:, :25] (I + M Pq)^-1 M P[:25, :],   Pq = P[:25, :25],
//     evaluated through Pq = Lc Lc^T and the SPD matrix B = I + Lc^T M Lc (push-through identity) - identical
//     to the reference's formulas in exact arithmetic;  diag(S) for the gate is the row norms of Hq Lc, + R;
//   * every product is 16 x 16 fp64 MFMA tiles (tile_gemm); the two 25 x 25 Cholesky factorisations run in ONE wave's
//     registers (wave_chol32: the 16 x 16 register factorisation of dense80.hpp twice + MFMA glue), chol(Pq) in
//     wave 3 while waves 0-2 evaluate the 572 sines / cosines; triangular solves are products with U = L^-T.
// The smoother gains A_i = P_est[i] F^T inv(P_pred[i+1]) are independent across frames (one workgroup each, P_pred
// rebuilt from P_est, blocked Cholesky + two triangular products, pivoting fallback); only the 75x75 mat-vec
// recursion is sequential (one workgroup per sequence).  The smoothed covariances are not computed: the reference
// never saves them (:850-857).
#include "cheetah_fk.hpp"
#include "dense80.hpp"

namespace acino {

constexpr int EP = 25;            // pose parameters (qb_list order)
constexpr int ES = 75;            // states: pose, velocity, acceleration
constexpr int EKF_MAXC = 6;
constexpr int EROWS = EKF_MAXC * 2 * NL;   // 240
constexpr int PLD = 81;           // leading dimension of the covariance in LDS: 80 x 81, zero beyond 75 (no masks on tiles)
constexpr int PR = 80;            // its padded row count
constexpr int VLD = 28;           // leading dimension of V = P[:, :25] A (80 x 28, zero beyond 75 x 25)
constexpr int HLD = 27;           // leading dimension of Hq
constexpr int SLD = 27;           // leading dimension of the 25 x 25 (+1) work matrices
constexpr int NVAR = EP + 1;      // base pose + one forward-difference variant per parameter


struct FkLite {
  static constexpr bool kHasOm = false;
  double sc[22][2];
  double pos[21][3];
};

// EKF parameter (qb_list order, :734-746) -> active-state index of the kinematic chain
// (0-2 xyz, 3-5 phi0,phi1,phi3, 6-19 theta0-13, 20-24 psi0,1,3,4,5)
__device__ const int8_t c_ekf2act[EP] = {0, 1, 2, 3, 6, 20, 4, 7, 21, 8, 5, 9, 22, 10, 23, 11, 24, 12, 13, 14, 15, 16, 17, 18, 19};
__device__ const double c_qb_list[EP] = {5.0,  5.0,  5.0,   10.0,  10.0,  10.0,  5.0,   25.0,  5.0,   50.0,  5.0,   50.0, 25.0,
                                         100.0, 30.0, 140.0, 40.0, 350.0, 200.0, 350.0, 200.0, 450.0, 400.0, 450.0, 400.0};

struct EkfK {
  int n_frames, n_cams, pivoting;
  double sT, dlc_thresh, max_pixel_err, eps;
};

__device__ __forceinline__ double ekf_readlane(double x, int lane) {
  long long b = __builtin_bit_cast(long long, x);
  int lo = __builtin_amdgcn_readlane((int)b, lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// Cholesky of the 25 x 25 SPD matrix A (LDS, leading dimension lda, lower triangle read) by ONE wave, blocked 16 + 9
// (the second tile padded with identity): the register-resident 16 x 16 factorisation of dense80.hpp twice, the panel
// L10 = A10 U00 and the Schur complement A11 - L10 L10^T on the matrix cores in between.  Lw (32 x 33) receives L
// (tiles (0,0), (1,0), (1,1)); Uw (32 x 33) the diagonal inverse factors U00 = L00^-T, U11 = L11^-T and, with WITH_U,
// the whole U = L^-T = [[U00, -U00 L10^T U11], [0, U11]] - a triangular solve then is a product with U^T.
// Everything a lane reads back was written by its own wave (LDS operations of one wave complete in order).
constexpr int WLD = 33;
template <bool WITH_U>
__device__ bool wave_chol32(const double* A, int lda, double* Lw, double* Uw, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  d4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = lk + 4 * r;
    acc[r] = A[(i > li ? i : li) * lda + (i > li ? li : i)];
  }
  bool ok = chol16_inv_acc<WLD, true>(Uw, acc, lane, nullptr, Lw);
  double av[4], bv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {                 // panel L10 = A10 U00
    av[s] = (16 + li < EP) ? A[(16 + li) * lda + 4 * s + lk] : 0.0;
    bv[s] = Uw[(4 * s + lk) * WLD + li];
  }
  d4 l10 = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; ++s) l10 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], l10, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Lw[(16 + lk + 4 * r) * WLD + li] = l10[r];
#pragma unroll
  for (int r = 0; r < 4; ++r) {                 // A11 (identity on the padding)
    const int i = 16 + lk + 4 * r, j = 16 + li;
    acc[r] = (i < EP && j < EP) ? A[(i > j ? i : j) * lda + (i > j ? j : i)] : (i == j ? 1.0 : 0.0);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) av[s] = Lw[(16 + li) * WLD + 4 * s + lk];
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[s], av[s], acc, 0, 0, 0);
  ok = chol16_inv_acc<WLD, true>(Uw + 16 * WLD + 16, acc, lane, nullptr, Lw + 16 * WLD + 16) && ok;
  if (WITH_U) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {               // Y = L10^T U11
      av[s] = Lw[(16 + 4 * s + lk) * WLD + li];
      bv[s] = Uw[(16 + 4 * s + lk) * WLD + 16 + li];
    }
    d4 y = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], y, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Uw[(lk + 4 * r) * WLD + 16 + li] = y[r];
      Uw[(16 + lk + 4 * r) * WLD + li] = 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {               // U01 = -U00 Y
      av[s] = Uw[li * WLD + 4 * s + lk];
      bv[s] = Uw[(4 * s + lk) * WLD + 16 + li];
    }
    d4 u01 = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) u01 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[s], bv[s], u01, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Uw[(lk + 4 * r) * WLD + 16 + li] = u01[r];
  }
  return ok;
}

// C = A B on the matrix cores with 16 x 16 output tiles dealt to the four waves; KS k-steps of 4.  The operands and
// the result go through accessors (a_at(row, k), b_at(k, col), c_out(row, col, value)) that mask the padding.
template <int KS, class FA, class FB, class FC>
__device__ __forceinline__ void tile_gemm(int tiles_m, int tiles_n, int wave, int li, int lk, FA a_at, FB b_at, FC c_out) {
  for (int t = wave; t < tiles_m * tiles_n; t += 4) {
    const int ti = t / tiles_n, tj = t % tiles_n;
    double av[KS], bv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      av[s] = a_at(16 * ti + li, 4 * s + lk);
      bv[s] = b_at(4 * s + lk, 16 * tj + li);
    }
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c_out(16 * ti + lk + 4 * r, 16 * tj + li, acc[r]);
  }
}

// Entry (i, j) of each of the nine 25 x 25 blocks of F P F^T + Q (:761-766, :733-757), F = [[I, a1 I, a2 I], [0, I, a1 I],
// [0, 0, I]].  Reads and writes the same nine positions, so src == dst is fine with one thread per (i, j).  The
// smoother's gain kernel rebuilds P_pred[i+1] from P_est[i] with this same code - the filter does not store it.
__device__ __forceinline__ void predict_cov_entry(const double* src, int lds, double* dst, int ldd, int i, int j, double sT) {
  const double a1 = sT, a2 = sT * sT / 2;
  double p[3][3], r[3][3];
#pragma unroll
  for (int bi = 0; bi < 3; ++bi)
#pragma unroll
    for (int bj = 0; bj < 3; ++bj) p[bi][bj] = src[(bi * EP + i) * lds + bj * EP + j];
#pragma unroll
  for (int bj = 0; bj < 3; ++bj) {
    r[0][bj] = p[0][bj] + a1 * p[1][bj] + a2 * p[2][bj];
    r[1][bj] = p[1][bj] + a1 * p[2][bj];
    r[2][bj] = p[2][bj];
  }
  double qd = 0.0;
  if (i == j) {
    const double h = c_qb_list[i] / 2;
    qd = h * h;
  }
  const double s2 = sT * sT;
  const double qc[3][3] = {{s2 * s2 / 4, s2 * sT / 2, s2 / 2}, {s2 * sT / 2, s2, sT}, {s2 / 2, sT, 1.0}};
#pragma unroll
  for (int bi = 0; bi < 3; ++bi) {
    dst[(bi * EP + i) * ldd + j] = r[bi][0] + a1 * r[bi][1] + a2 * r[bi][2] + qc[bi][0] * qd;
    dst[(bi * EP + i) * ldd + EP + j] = r[bi][1] + a1 * r[bi][2] + qc[bi][1] * qd;
    dst[(bi * EP + i) * ldd + 2 * EP + j] = r[bi][2] + qc[bi][2] * qd;
  }
}

__global__ void __launch_bounds__(256)
k_ekf_forward(EkfK K, const double* __restrict__ det_all, const double* __restrict__ cams,
              const double* __restrict__ states0_all, double* __restrict__ x_pred_all, double* __restrict__ x_est_all,
              double* __restrict__ P_est_all, int* __restrict__ outliers,
              int* __restrict__ numeric_err) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* P = sm;                       // [75][76]
  double* Hq = P + PR * PLD;            // [240][27]   later: V [80][28] and Ptop [28][81]
  double* hv = Hq + EROWS * HLD;        // [240] h(x)
  double* rs = hv + EROWS;              // [240] residual
  double* ri = rs + EROWS;              // [240] 1 / R
  double* sd = ri + EROWS;              // [240] diag S
  double* xs = sd + EROWS;              // [80]
  double* gv = xs + 80;                 // [32]
  double* tv = gv + 32;                 // [32]
  double* scr = tv + 32;                // FK frames, then the 25 x 27 work matrices
  Cam* cm = reinterpret_cast<Cam*>(scr + 2800 + 2 * 32 * WLD);
  unsigned* dep = reinterpret_cast<unsigned*>(cm + EKF_MAXC);          // [20]
  int* n_pairs = reinterpret_cast<int*>(dep + NL);
  unsigned short* pairs = reinterpret_cast<unsigned short*>(n_pairs + 4);    // [<= 20 + 25 * 20]
  FkLite* fr = reinterpret_cast<FkLite*>(scr);
  double* Mm = scr;                     // M (25 x 26: column 25 = g)
  double* N1 = Mm + EP * SLD;           // Lc^T [M | g]
  double* Bm = N1 + EP * SLD;           // B -> Lb, then Z = Lb^-1 N1
  double* Am = Bm + EP * SLD;           // A
  double* Lc = scr + 2800;              // chol(Pq) [32][33]; beyond the FK frames (26 x 107 doubles): built early
  double* Ub = Lc + 32 * WLD;           // [32][33] inverse factors: scratch of chol(Pq), then U = Lb^-T
  double* V = Hq;                       // [75][27]
  double* Pt = Hq + PR * VLD;           // [28][81]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int seq = blockIdx.x, N = K.n_frames, C = K.n_cams, rows = C * 2 * NL;
  const double* det = det_all + (size_t)seq * N * C * NL * 3;
  double* x_pred = x_pred_all + (size_t)seq * N * ES;
  double* x_est = x_est_all + (size_t)seq * N * ES;
  double* P_est = P_est_all + (size_t)seq * N * ES * ES;
  const double sT = K.sT, a1 = sT, a2 = sT * sT / 2;

  for (int e = tid; e < C * ACINO_CAM_STRIDE; e += 256) reinterpret_cast<double*>(cm)[e] = cams[e];
  for (int e = tid; e < PR * PLD; e += 256) P[e] = 0.0;
  if (tid < ES) xs[tid] = states0_all[(size_t)seq * ES + tid];
  __syncthreads();
  if (tid < ES) {   // P0 (:713-730)
    const int blk = tid / EP, p = tid % EP;
    double v;
    if (blk == 0) v = p < 3 ? 9.0 : (M_PI / 4) * (M_PI / 4);
    else if (blk == 1) v = p < 3 ? 25.0 : 9.0;
    else v = (p >= 3 + 10) ? 25.0 : 9.0;
    P[tid * PLD + tid] = v;
  }
  int n_out = 0;
  bool bad = false;
  int bad_code = 0;
  __syncthreads();

  // sines / cosines and head positions of the 26 pose variants of ``pose`` (threads tid < nthr, stride nthr)
  auto fill_frames = [&](const double* pose, int nthr) {
    for (int task = tid; task < NVAR * 22; task += nthr) {
      const int v = task / 22, a = task % 22;               // active angle a + 3
      int p = 0;
#pragma unroll
      for (int q = 3; q < EP; ++q) p = (c_ekf2act[q] == a + 3) ? q : p;
      // the reference perturbs its FLOAT32 state array (:628, :640): fl32(x_p) + fl32(eps), a float32 sum, while the
      // difference quotient divides by the float64 eps (:643) - the filter's output carries that ~1e-4 column scaling
      const double ang = (v - 1 == p) ? (double)((float)pose[p] + (float)K.eps) : pose[p];
      double s, c;
      sincos(ang, &s, &c);
      fr[v].sc[a][0] = s;
      fr[v].sc[a][1] = c;
    }
    for (int task = tid; task < NVAR * 3; task += nthr) {
      const int v = task / 3, c = task % 3;
      fr[v].pos[20][c] = (v - 1 == c) ? (double)((float)pose[c] + (float)K.eps) : pose[c];
    }
  };
  // Which marker moves with which parameter is a property of the kinematic chain: probe it once at a generic pose
  // (a marker whose position is bit-identical in the perturbed frame gives a forward difference of exactly 0 at every
  // pose - those projections are skipped and the Jacobian entry is written as 0).  pairs[] = the base variant's 20
  // markers followed by the dependent (variant, marker) pairs; dep[l] = bit mask over the parameters.
  {
    if (tid < EP) hv[tid] = 0.3 + 0.17 * tid;
    __syncthreads();
    fill_frames(hv, 256);
    __syncthreads();
    if (tid < NVAR * 3) fk_columns(fr[tid / 3], tid % 3);
    __syncthreads();
    unsigned char* flag = reinterpret_cast<unsigned char*>(Hq);
    for (int e = tid; e < EP * NL; e += 256) {
      const int v = 1 + e / NL, l = e % NL;
      flag[e] = fr[v].pos[l][0] != fr[0].pos[l][0] || fr[v].pos[l][1] != fr[0].pos[l][1] || fr[v].pos[l][2] != fr[0].pos[l][2];
    }
    __syncthreads();
    if (tid < NL) {
      unsigned m = 0;
      for (int p = 0; p < EP; ++p) m |= flag[p * NL + tid] ? (1u << p) : 0u;
      dep[tid] = m;
      pairs[tid] = (unsigned short)tid;
    }
    if (tid == 0) {
      int n = NL;
      for (int e = 0; e < EP * NL; ++e)
        if (flag[e]) pairs[n++] = (unsigned short)(((1 + e / NL) << 5) | (e % NL));
      *n_pairs = n;
    }
    __syncthreads();
  }
  const int np = *n_pairs;

  for (int f = 0; f < N; ++f) {
    // ---- predict (:622-628; the reference rounds the predicted state to float32) ----
    if (tid < EP) {
      const double acc = xs[2 * EP + tid];
      const double vel = __dadd_rn(xs[EP + tid], __dmul_rn(sT, acc));
      const double pos = __dadd_rn(__dadd_rn(xs[tid], __dmul_rn(sT, vel)), __dmul_rn(__dmul_rn(0.5, __dmul_rn(sT, sT)), acc));
      xs[tid] = (double)(float)pos;
      xs[EP + tid] = (double)(float)vel;
      xs[2 * EP + tid] = (double)(float)acc;
    }
    // P <- F P F^T + Q, block-wise: F = [[I, a1 I, a2 I], [0, I, a1 I], [0, 0, I]]
    for (int e = tid; e < EP * EP; e += 256) predict_cov_entry(P, PLD, P, PLD, e / EP, e % EP, sT);
    __syncthreads();
    if (tid < ES) x_pred[(size_t)f * ES + tid] = xs[tid];

    // ---- measurement model: 26 pose variants (base + eps on each parameter) ----
    // waves 0-2 build the kinematic frames; wave 3 meanwhile factors Pq = Lc Lc^T in its registers
    if (wave < 3) {
      fill_frames(xs, 192);
      for (int e = tid; e < rows * EP; e += 192) Hq[(e / EP) * HLD + e % EP] = 0.0;     // entries of markers that do not move
    } else {
      if (!wave_chol32<false>(P, PLD, Lc, Ub, lane) && !bad) { bad = true; bad_code = 2 * f + 1; }
    }
    __syncthreads();
    if (tid < NVAR * 3) fk_columns(fr[tid / 3], tid % 3);
    __syncthreads();
    for (int task = tid; task < np * C; task += 256) {
      const int pr = pairs[task / C], c = task % C, v = pr >> 5, l = pr & 31;
      double u, w;
      project_fisheye_pt(cm[c], fr[v].pos[l][0], fr[v].pos[l][1], fr[v].pos[l][2], u, w);
      const int row = c * 2 * NL + 2 * l;
      if (v == 0) {
        hv[row] = u;
        hv[row + 1] = w;
      } else {
        Hq[row * HLD + v - 1] = u;
        Hq[(row + 1) * HLD + v - 1] = w;
      }
    }
    __syncthreads();
    for (int task = tid; task < (np - NL) * C * 2; task += 256) {
      const int pr = pairs[NL + task / (2 * C)], c = (task >> 1) % C, row = c * 2 * NL + 2 * (pr & 31) + (task & 1);
      double* h = Hq + row * HLD + (pr >> 5) - 1;
      *h = (*h - hv[row]) / K.eps;       // (:643)
    }
    if (tid < rows) {
      const int c = tid / (2 * NL), l = (tid % (2 * NL)) / 2, d = tid & 1;
      const double* dd = det + (((size_t)f * C + c) * NL + l) * 3;
      rs[tid] = dd[d] - hv[tid];
      const double sdv = dd[2] < K.dlc_thresh ? K.max_pixel_err : 25.0;     // (:805-809: the 5**2 is squared again)
      ri[tid] = 1.0 / (sdv * sdv);
    }
    __syncthreads();
    // diag S = Hq Pq Hq^T + R = row norms of Hq Lc, + R: G^T = Lc^T Hq^T on the matrix cores, 16 measurement rows per
    // tile column (lane li = row, the 25 entries of a row spread over 2 tiles x 4 registers x 4 lane groups)
    for (int tr = wave; tr < (rows + 15) / 16; tr += 4) {     // (rows = 40 C: the last tile is half empty for odd C)
      const int row = 16 * tr + li;
      double hq[7], lc0[7], lc1[7];
#pragma unroll
      for (int s7 = 0; s7 < 7; ++s7) {
        const int kk = 4 * s7 + lk;
        hq[s7] = kk < EP ? Hq[row * HLD + kk] : 0.0;
        lc0[s7] = (kk < EP && kk >= li) ? Lc[kk * WLD + li] : 0.0;
        lc1[s7] = (kk < EP && li + 16 < EP && kk >= li + 16) ? Lc[kk * WLD + li + 16] : 0.0;
      }
      d4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
#pragma unroll
      for (int s7 = 0; s7 < 7; ++s7) {
        g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lc0[s7], hq[s7], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lc1[s7], hq[s7], g1, 0, 0, 0);
      }
      double ss = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) ss += g0[r] * g0[r] + g1[r] * g1[r];
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lk == 0 && row < rows) sd[row] = ss + 1.0 / ri[row];
    }
    __syncthreads();
    if (tid < rows / 2) {   // 3-sigma gate per pixel pair (:813-819)
      const int j = 2 * tid;
      if (fabs(rs[j]) > 3.0 * sqrt(sd[j]) || fabs(rs[j + 1]) > 3.0 * sqrt(sd[j + 1])) {
        rs[j] = 0.0;
        rs[j + 1] = 0.0;
        ++n_out;
      }
    }
    __syncthreads();
    // [M | g] = Hq^T R^-1 [Hq | r] on the matrix cores: 32 x 32 padded, one 16 x 16 tile per wave
    {
      const int ti = wave >> 1, tj = wave & 1;
      const int arow = 16 * ti + li, bcol = 16 * tj + li;
      d4 acc = {0, 0, 0, 0};
      const bool a_on = arow < EP, b_h = bcol < EP, b_r = bcol == EP;
      const double* pa = Hq + (a_on ? arow : 0);
      const double* pb = b_h ? Hq + bcol : rs;            // column of Hq, or the residual as the 26th column
      const int sb = b_h ? HLD : 1;
      for (int s = 0; s < rows / 4; s += 5) {             // rows / 4 = 10 C; the operands of five steps in flight
        double av[5], bw[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int k0 = 4 * (s + u) + lk;
          av[u] = pa[k0 * HLD];
          bw[u] = ri[k0] * pb[k0 * sb];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_on ? av[u] : 0.0, (b_h || b_r) ? bw[u] : 0.0, acc, 0, 0, 0);
      }
      // (the FK frames that share scr were last read by the projections, several barriers ago)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
        if (row < EP && col <= EP) Mm[row * SLD + col] = acc[r];
      }
    }
    __syncthreads();
    // N1 = Lc^T [M | g]
    tile_gemm<7>(2, 2, wave, li, lk,
                 [&](int a_, int k) { return (a_ < EP && k < EP && k >= a_) ? Lc[k * WLD + a_] : 0.0; },
                 [&](int k, int b_) { return (k < EP && b_ <= EP) ? Mm[k * SLD + b_] : 0.0; },
                 [&](int a_, int b_, double v) { if (a_ < EP && b_ <= EP) N1[a_ * SLD + b_] = v; });
    __syncthreads();
    // B = I + N1 Lc
    tile_gemm<7>(2, 2, wave, li, lk,
                 [&](int a_, int k) { return (a_ < EP && k < EP) ? N1[a_ * SLD + k] : 0.0; },
                 [&](int k, int b_) { return (k < EP && b_ < EP && k >= b_) ? Lc[k * WLD + b_] : 0.0; },
                 [&](int a_, int b_, double v) { if (a_ < EP && b_ < EP) Bm[a_ * SLD + b_] = v + (a_ == b_ ? 1.0 : 0.0); });
    __syncthreads();
    // B = Lb Lb^T with U = Lb^-T alongside, in one wave's registers
    // (waves 1-3 meanwhile copy the top rows of P for the covariance update: the Jacobian storage they go to is free)
    if (wave == 0) {
      if (!wave_chol32<true>(Bm, SLD, Lc, Ub, lane) && !bad) { bad = true; bad_code = 2 * f + 2; }   // (Lc is done with)
    } else {
      for (int e = tid - 64; e < 28 * PLD; e += 192) Pt[e] = e < EP * PLD ? P[e] : 0.0;
    }
    __syncthreads();
    // Z = Lb^-1 [N1 | u] = U^T [N1 | u]   (into Bm: Lb itself is not needed again)
    tile_gemm<7>(2, 2, wave, li, lk,
                 [&](int a_, int k) { return (a_ < EP && k <= a_) ? Ub[k * WLD + a_] : 0.0; },
                 [&](int k, int b_) { return (k < EP && b_ <= EP) ? N1[k * SLD + b_] : 0.0; },
                 [&](int a_, int b_, double v) { if (a_ < EP && b_ <= EP) Bm[a_ * SLD + b_] = v; });
    __syncthreads();
    // A = M - Z^T Z ;  w = g - Z^T zu
    tile_gemm<7>(2, 2, wave, li, lk,
                 [&](int a_, int k) { return (a_ < EP && k < EP) ? Bm[k * SLD + a_] : 0.0; },
                 [&](int k, int b_) { return (k < EP && b_ <= EP) ? Bm[k * SLD + b_] : 0.0; },
                 [&](int a_, int b_, double v) {
                   if (a_ < EP && b_ < EP) Am[a_ * SLD + b_] = Mm[a_ * SLD + b_] - v;
                   else if (a_ < EP && b_ == EP) gv[a_] = Mm[a_ * SLD + EP] - v;
                 });
    __syncthreads();
    // state correction x += P[:, :25] w ; V = P[:, :25] A ; Ptop = P[:25, :]
    if (tid < ES) {
      double s = 0.0;
      for (int a = 0; a < EP; ++a) s += P[tid * PLD + a] * gv[a];
      xs[tid] += s;
    }
    tile_gemm<7>(5, 2, wave, li, lk,
                 [&](int r_, int k) { return k < EP ? P[r_ * PLD + k] : 0.0; },
                 [&](int k, int b_) { return (k < EP && b_ < EP) ? Am[k * SLD + b_] : 0.0; },
                 [&](int r_, int b_, double v) { if (b_ < VLD) V[r_ * VLD + b_] = v; });     // zero beyond 75 x 25
    __syncthreads();
    // P -= V Ptop on the matrix cores: 5 x 5 tiles of the 75 x 75 matrix (padded to 80), K = 25 padded to 28
    for (int q0 = 0; q0 < 7; q0 += 2) {      // two tiles' operands in flight per wave
      d4 acc[2];
      double av[2][7], bv[2][7];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = wave + 4 * (q0 + q);
        if (t < 25) {
          const int ti = t / 5, tj = t % 5;
          const int rowA = 16 * ti + li, colB = 16 * tj + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + lk + 4 * r;
            acc[q][r] = P[row * PLD + colB];
          }
#pragma unroll
          for (int s7 = 0; s7 < 7; ++s7) {
            const int kk = 4 * s7 + lk;
            av[q][s7] = V[rowA * VLD + kk];
            bv[q][s7] = Pt[kk * PLD + colB];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = wave + 4 * (q0 + q);
        if (t < 25) {
#pragma unroll
          for (int s7 = 0; s7 < 7; ++s7) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[q][s7], bv[q][s7], acc[q], 0, 0, 0);
          const int ti = t / 5, tj = t % 5, colB = 16 * tj + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + lk + 4 * r;
            P[row * PLD + colB] = acc[q][r];
            if (row < ES && colB < ES) P_est[(size_t)f * ES * ES + row * ES + colB] = acc[q][r];   // the smoother's input
          }
        }
      }
    }
    if (tid < ES) x_est[(size_t)f * ES + tid] = xs[tid];
    __syncthreads();
  }
  // outlier count: one pair per thread per frame
  for (int off = 32; off > 0; off >>= 1) n_out += __shfl_down(n_out, off, 64);
  if (lane == 0 && n_out) atomicAdd(&outliers[seq], n_out);
  if (bad && lane == 0) atomicCAS(numeric_err, 0, bad_code);   // first failure: 2 f + 1 (P) or 2 f + 2 (B)
}

// A_i = P_est[i] F^T inv(P_pred[i+1]) for i = 1 .. N-2 (one workgroup per (sequence, frame)), P_pred[i+1] = F P_est[i] F^T + Q
// rebuilt here;  A_i^T = inv(P_pred[i+1]) G, G = F P_est[i]^T.  P_pred is symmetric positive definite while the filter
// is healthy: blocked Cholesky with the inverse factor U = L^-T (dense80.hpp), then A_i = (U^T G)^T U^T - two
// triangular 80 x 80 products on the matrix cores.  If a pivot is not positive ((I - K H) P does not keep positive
// definiteness to round-off when a filter has lost its target), the workgroup falls back to Gauss-Jordan elimination
// with partial pivoting on [P_pred | G] - like the reference's np.linalg.inv (:840), no definiteness requirement.
__device__ void rts_build(double* Lp, double* G, const double* E, double sT, int tid) {
  const double a1 = sT, a2 = sT * sT / 2;
  for (int e = tid; e < EP * EP; e += 256) predict_cov_entry(E, LD, Lp, LD, e / EP, e % EP, sT);
  for (int e = tid; e < BS * BS; e += 256) {
    const int r = e / BS, c = e % BS;
    double g = 0.0;
    if (r < ES && c < ES) {     // G[r][c] = (F P_est^T)[r][c] = sum_k F[r][k] P_est[c][k]
      g = E[c * LD + r];
      if (r < 2 * EP) g += a1 * E[c * LD + r + EP];
      if (r < EP) g += a2 * E[c * LD + r + 2 * EP];
    } else {
      Lp[r * LD + c] = (r == c) ? 1.0 : 0.0;
    }
    G[r * LD + c] = g;
  }
}

__global__ void __launch_bounds__(256)
k_rts_gain(EkfK K, const double* __restrict__ P_est_all, double* __restrict__ A_all, int* __restrict__ numeric_err) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Lp = sm;                 // [80][81] P_pred -> L / U
  double* G = Lp + MAT;            // [80][81] F P_est^T -> U^T G
  double* E = G + MAT;             // [80][81] P_est
  __shared__ int s_err, s_piv;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int N = K.n_frames, per = N - 2;
  const int seq = blockIdx.x / per, i = 1 + blockIdx.x % per;
  const double* Pe = P_est_all + ((size_t)seq * N + i) * ES * ES;
  double* At = A_all + ((size_t)seq * N + i) * ES * ES;     // stored TRANSPOSED: the recursion reads columns
  for (int e = tid; e < ES * ES; e += 256) E[(e / ES) * LD + e % ES] = Pe[e];
  if (tid == 0) s_err = 0;
  __syncthreads();
  rts_build(Lp, G, E, K.sT, tid);
  __syncthreads();
  if (!K.pivoting) chol80(Lp, tid, &s_err);
  if (!K.pivoting && !s_err) {
    d4 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {          // T = U^T G : tile (ti, tj) = sum_{k <= ti} U(k, ti)^T G(k, tj)
      const int t = wave + 4 * q;
      acc[q] = d4{0, 0, 0, 0};
      if (t < 25) {
        const int ti = t / 5, tj = t % 5;
        for (int k = 0; k <= ti; ++k)
          acc[q] = mma_seq<4, false>(acc[q], Lp + (16 * k + lk) * LD + 16 * ti + li, 4 * LD, G + (16 * k + lk) * LD + 16 * tj + li,
                                     4 * LD);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int t = wave + 4 * q;
      if (t < 25) {
        const int ti = t / 5, tj = t % 5;
#pragma unroll
        for (int r = 0; r < 4; ++r) G[(16 * ti + lk + 4 * r) * LD + 16 * tj + li] = acc[q][r];
      }
    }
    __syncthreads();
    for (int t = wave; t < 25; t += 4) {   // A^T = U T : tile (ti, tj) = sum_{k >= ti} U(ti, k) T(k, tj)
      const int ti = t / 5, tj = t % 5;
      d4 a = {0, 0, 0, 0};
      for (int k = ti; k < 5; ++k)
        a = mma_seq<4, false>(a, Lp + (16 * ti + li) * LD + 16 * k + lk, 4, G + (16 * k + lk) * LD + 16 * tj + li, 4 * LD);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
        if (row < ES && col < ES) At[row * ES + col] = a[r];
      }
    }
    return;
  }
  __syncthreads();
  rts_build(Lp, G, E, K.sT, tid);
  __syncthreads();
  for (int k = 0; k < ES; ++k) {
    if (tid < 64) {
      double best = -1.0;
      int bi = k;
      for (int r = k + lane; r < ES; r += 64) {
        const double v = fabs(Lp[r * LD + k]);
        if (v > best) {
          best = v;
          bi = r;
        }
      }
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) {
          best = ob;
          bi = oi;
        }
      }
      if (lane == 0) {
        s_piv = bi;
        if (!(best > 0.0)) atomicCAS(numeric_err, 0, -(i + 1));
      }
    }
    __syncthreads();
    const int pr = s_piv;
    if (pr != k && tid < 2 * ES) {
      double* M2 = tid < ES ? Lp : G;
      const int c = tid % ES;
      const double t = M2[k * LD + c];
      M2[k * LD + c] = M2[pr * LD + c];
      M2[pr * LD + c] = t;
    }
    __syncthreads();
    const double d = Lp[k * LD + k];
    __syncthreads();
    const double inv = 1.0 / (d != 0.0 ? d : 1.0);
    if (tid < 2 * ES) {
      double* M2 = tid < ES ? Lp : G;
      M2[k * LD + tid % ES] *= inv;
    }
    __syncthreads();
    const int nc = (ES - k - 1) + ES;          // columns k+1.. of P_pred, all of G
    for (int e = tid; e < ES * nc; e += 256) {
      const int r = e / nc, cc = e % nc;
      if (r == k) continue;
      const double fct = Lp[r * LD + k];
      if (cc < ES - k - 1) Lp[r * LD + k + 1 + cc] -= fct * Lp[k * LD + k + 1 + cc];
      else G[r * LD + cc - (ES - k - 1)] -= fct * G[k * LD + cc - (ES - k - 1)];
    }
    __syncthreads();
  }
  for (int e = tid; e < ES * ES; e += 256) At[e] = G[(e / ES) * LD + e % ES];     // A^T = X
}

// smooth[i] = x_est[i] + A_i (smooth[i+1] - x_pred[i+1]), i = N-2 .. 1; frames 0 and N-1 keep the filtered state (:836-839).
// The recursion is a chain of 75 x 75 mat-vecs: one thread per row keeps its row of A_i in registers (the next
// frame's row is already in flight), the difference vector is double-buffered in LDS - ONE barrier per frame.
__global__ void __launch_bounds__(128)
k_rts_recurse(EkfK K, const double* __restrict__ x_pred_all, const double* __restrict__ x_est_all,
              const double* __restrict__ A_all, double* __restrict__ smooth_all) {
  __shared__ __attribute__((aligned(16))) double v[2][ES + 1];
  const int tid = threadIdx.x, N = K.n_frames, seq = blockIdx.x;
  const double* x_pred = x_pred_all + (size_t)seq * N * ES;
  const double* x_est = x_est_all + (size_t)seq * N * ES;
  const double* A = A_all + (size_t)seq * N * ES * ES;
  double* smooth = smooth_all + (size_t)seq * N * ES;
  const bool on = tid < ES;
  if (on) {
    smooth[tid] = x_est[tid];
    if (N > 1) smooth[(size_t)(N - 1) * ES + tid] = x_est[(size_t)(N - 1) * ES + tid];
  }
  if (N < 3) return;
  double a[ES], an[ES];
  double xe = 0.0, xp_next = 0.0;
  if (on) {
    v[0][tid] = x_est[(size_t)(N - 1) * ES + tid] - x_pred[(size_t)(N - 1) * ES + tid];
    xe = x_est[(size_t)(N - 2) * ES + tid];
    xp_next = x_pred[(size_t)(N - 2) * ES + tid];
#pragma unroll
    for (int c = 0; c < ES; ++c) a[c] = A[(size_t)(N - 2) * ES * ES + c * ES + tid];     // A is stored transposed
  }
  __syncthreads();
  int buf = 0;
  for (int i = N - 2; i >= 1; --i) {
    double xe_n = 0.0, xp_n = 0.0;
    if (on && i > 1) {                 // the next frame's operands: in flight during this frame's dot product
      xe_n = x_est[(size_t)(i - 1) * ES + tid];
      xp_n = x_pred[(size_t)(i - 1) * ES + tid];
#pragma unroll
      for (int c = 0; c < ES; ++c) an[c] = A[(size_t)(i - 1) * ES * ES + c * ES + tid];
    }
    if (on) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int c = 0; c < ES; c += 3) {
        s0 += a[c] * v[buf][c];
        s1 += a[c + 1] * v[buf][c + 1];
        s2 += a[c + 2] * v[buf][c + 2];
      }
      const double cur = xe + ((s0 + s1) + s2);
      smooth[(size_t)i * ES + tid] = cur;
      v[buf ^ 1][tid] = cur - xp_next;       // smooth[i] - x_pred[i]: the next frame's right-hand side
#pragma unroll
      for (int c = 0; c < ES; ++c) a[c] = an[c];
      xe = xe_n;
      xp_next = xp_n;
    }
    buf ^= 1;
    __syncthreads();
  }
}

static size_t a256(size_t v) { return (v + 255) / 256 * 256; }
constexpr size_t kRtsLds = 3 * MAT * sizeof(double);
constexpr size_t kEkfLds = (size_t)(PR * PLD + EROWS * HLD + 4 * EROWS + 80 + 32 + 32 + 2800 + 2 * 32 * 33) * sizeof(double) +
                           EKF_MAXC * sizeof(Cam) + (NL + 4) * 4 + (NL + EP * NL) * 2 + 8;
static_assert(sizeof(FkLite) * NVAR <= 6 * EP * SLD * sizeof(double), "FK frames must fit the scratch region");
static_assert((PR * VLD + 28 * PLD) <= EROWS * HLD, "V and Ptop alias the Jacobian storage");
static_assert(kEkfLds <= 160 * 1024, "EKF workgroup state must fit the 160 KB LDS");

}  // namespace acino

using namespace acino;

extern "C" {

size_t acino_sizeof_ekf_params(void) { return sizeof(acino_ekf_params); }

size_t acino_ekf_workspace_bytes(int64_t n_frames, int n_seq) {
  if (n_frames < 1 || n_seq < 1) return 0;
  const size_t N = (size_t)n_frames * n_seq;
  return 2 * a256(N * ES * ES * 8) + 2 * a256(N * ES * 8) + 1024;
}

int acino_ekf_run(const acino_ekf_params* prm, const double* d_det, const double* d_cams24, const double* d_states0,
                  void* d_ws, size_t ws_bytes, double* d_est, double* d_smooth, int32_t* d_outliers, void* stream) {
  ACINO_REQUIRE(prm, "params");
  ACINO_REQUIRE(prm->n_frames >= 1 && prm->n_seq >= 1, "n_frames, n_seq");
  ACINO_REQUIRE(prm->n_cams >= 1 && prm->n_cams <= EKF_MAXC, "n_cams in 1..6");
  ACINO_REQUIRE(prm->fps > 0 && prm->cam_width > 0, "fps, cam_width");
  ACINO_REQUIRE(d_det && d_cams24 && d_states0 && d_ws && d_est && d_smooth && d_outliers, "null buffer");
  ACINO_REQUIRE(((uintptr_t)d_ws & 255) == 0, "workspace must be 256-byte aligned");
  ACINO_REQUIRE(ws_bytes >= acino_ekf_workspace_bytes(prm->n_frames, prm->n_seq), "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t N = (size_t)prm->n_frames * prm->n_seq;
  char* w = (char*)d_ws;
  double* P_est = (double*)w;   w += a256(N * ES * ES * 8);
  double* A = (double*)w;       w += a256(N * ES * ES * 8);
  double* x_pred = (double*)w;  w += a256(N * ES * 8);
  int* nerr = (int*)w;
  EkfK K;
  K.n_frames = (int)prm->n_frames;
  K.n_cams = prm->n_cams;
  K.pivoting = prm->smoother_pivoting;
  K.sT = 1.0 / prm->fps;
  K.dlc_thresh = prm->dlc_thresh;
  K.max_pixel_err = prm->cam_width;
  K.eps = 1e-3;
  static PerDeviceOnce attr;
  if (attr.first()) {
    ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ekf_forward),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEkfLds));
    ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rts_gain),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRtsLds));
  }
  ACINO_HIP_CHECK(hipMemsetAsync(d_outliers, 0, sizeof(int32_t) * prm->n_seq, s));
  ACINO_HIP_CHECK(hipMemsetAsync(nerr, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_ekf_forward, dim3(prm->n_seq), dim3(256), kEkfLds, s, K, d_det, d_cams24, d_states0, x_pred, d_est,
                     P_est, d_outliers, nerr);
  ACINO_LAUNCH_CHECK();
  if (prm->n_frames >= 3) {
    hipLaunchKernelGGL(k_rts_gain, dim3((unsigned)(prm->n_seq * (prm->n_frames - 2))), dim3(256), kRtsLds, s, K, P_est, A, nerr);
    ACINO_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_rts_recurse, dim3(prm->n_seq), dim3(128), 0, s, K, x_pred, d_est, A, d_smooth);
  ACINO_LAUNCH_CHECK();
  int h_err = 0;
  ACINO_HIP_CHECK(hipMemcpyAsync(&h_err, nerr, sizeof(int), hipMemcpyDeviceToHost, s));
  ACINO_HIP_CHECK(hipStreamSynchronize(s));
  if (h_err) {
    if (h_err > 0)
      set_error("EKF: %s lost positive definiteness at frame %d", (h_err & 1) ? "the pose covariance" : "I + Lc^T M Lc",
                (h_err - 1) / 2);
    else
      set_error("EKF smoother: the predicted covariance of frame %d is singular", -h_err);
    return ACINO_ERR_NUMERIC;
  }
  return ACINO_OK;
}

}  // extern "C"
