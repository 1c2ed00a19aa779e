// Extended Kalman filter + Rauch-Tung-Striebel smoother over the 25 cheetah pose parameters (gfx950, fp64).
//
// Reference: `ekf` in src/all_optimizations.py:569-865 - constant-acceleration model (75 states), measurement =
// fisheye projection of the 20 markers in every camera (up to 240 pixel coordinates per frame), Jacobian by forward
// differences with eps = 1e-3 (:631-646), 3-sigma gating (:815-819), K = P H^T inv(S) with S 240x240 (:822),
// P <- (I - K H) P (:829), smoother (:836-841).  The filter is sequential in frames, so ONE workgroup runs a whole
// sequence (grid = sequences) with the 75x75 covariance, the 240x25 Jacobian and every intermediate in LDS:
//   * FK of the 26 forward-difference variants column-parallel (78 threads), 26 x C x 20 projections on all threads;
//   * H has 25 non-zero columns and R is diagonal, so with M = Hq^T R^-1 Hq, g = Hq^T R^-1 r (fp64 MFMA, g rides
//     along as a 26th column) the 240x240 inverse collapses to 25x25 algebra:
//       K r = P[:, :25] (I + M Pq)^-1 g,   P <- P - P[:, :25] (I + M Pq)^-1 M P[:25, :],   Pq = P[:25, :25],
//     evaluated through Pq = Lc Lc^T and the SPD matrix B = I + Lc^T M Lc (push-through identity) - identical
//     to the reference's formulas in exact arithmetic;  diag(S) for the gate is the row-wise form Hq Pq Hq^T + R.
// The smoother gains A_i = P_est[i] F^T inv(P_pred[i+1]) are independent across frames (one workgroup each); only
// the 75x75 mat-vec recursion is sequential (one workgroup per sequence).  The smoothed covariances are not computed:
// the reference never saves them (:850-857).
#include "cheetah_fk.hpp"

namespace acino {

constexpr int EP = 25;            // pose parameters (qb_list order)
constexpr int ES = 75;            // states: pose, velocity, acceleration
constexpr int EKF_MAXC = 6;
constexpr int EROWS = EKF_MAXC * 2 * NL;   // 240
constexpr int PLD = 76;           // leading dimension of 75 x 75 matrices in LDS
constexpr int HLD = 27;           // leading dimension of Hq
constexpr int SLD = 27;           // leading dimension of the 25 x 25 (+1) work matrices
constexpr int NVAR = EP + 1;      // base pose + one forward-difference variant per parameter

typedef double d4 __attribute__((ext_vector_type(4)));

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
  int n_frames, n_cams;
  double sT, dlc_thresh, max_pixel_err, eps;
};

__device__ __forceinline__ double ekf_readlane(double x, int lane) {
  long long b = __builtin_bit_cast(long long, x);
  int lo = __builtin_amdgcn_readlane((int)b, lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// Cholesky of the 25 x 25 SPD matrix A (LDS, leading dimension SLD) by ONE wave, in registers: lane i holds row i;
// pivots and multipliers are wave-uniform SGPR broadcasts (v_readlane), no barriers.  With NRHS > 0 the same wave
// then solves L Z = Rhs for the NRHS (<= 26) columns of Rhs (LDS, leading dimension SLD; lane c owns column c, the
// entries of L are broadcast from the lane that holds their row) and overwrites Rhs with Z.  The lower triangle of
// A is overwritten by L.  Returns false on a non-positive pivot.  Call from one wave; other waves wait at a barrier.
template <int NRHS>
__device__ bool wave_chol25(double* A, double* Rhs, int lane) {
  double row[EP];
#pragma unroll
  for (int j = 0; j < EP; ++j) row[j] = (lane < EP && j <= lane) ? A[lane * SLD + j] : 0.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < EP; ++k) {
    const double piv0 = ekf_readlane(row[k], k);
    ok = ok && (piv0 > 0.0);
    const double piv = fmax(piv0, 1e-300);
    double y = __builtin_amdgcn_rsq(piv);
    const double e = fma(-(piv * y), y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);
    const double lik = row[k] * y;                      // L[i][k] for lanes i >= k
    row[k] = lik;
#pragma unroll
    for (int j = k + 1; j < EP; ++j) row[j] -= lik * ekf_readlane(lik, j);
  }
  if (lane < EP) {
#pragma unroll
    for (int j = 0; j < EP; ++j)
      if (j <= lane) A[lane * SLD + j] = row[j];
  }
  if (NRHS > 0) {
    double z[EP];
#pragma unroll
    for (int k = 0; k < EP; ++k) z[k] = lane < NRHS ? Rhs[k * SLD + lane] : 0.0;
#pragma unroll
    for (int k = 0; k < EP; ++k) {
      double sacc = z[k];
#pragma unroll
      for (int j = 0; j < k; ++j) sacc -= ekf_readlane(row[j], k) * z[j];     // L[k][j] lives in lane k
      z[k] = sacc / ekf_readlane(row[k], k);
    }
    if (lane < NRHS) {
#pragma unroll
      for (int k = 0; k < EP; ++k) Rhs[k * SLD + lane] = z[k];
    }
  }
  return ok;
}

__global__ void __launch_bounds__(256)
k_ekf_forward(EkfK K, const double* __restrict__ det_all, const double* __restrict__ cams,
              const double* __restrict__ states0_all, double* __restrict__ x_pred_all, double* __restrict__ x_est_all,
              double* __restrict__ P_pred_all, double* __restrict__ P_est_all, int* __restrict__ outliers,
              int* __restrict__ numeric_err) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* P = sm;                       // [75][76]
  double* Hq = P + ES * PLD;            // [240][27]   later: V [75][27] and Ptop [25][76]
  double* hv = Hq + EROWS * HLD;        // [240] h(x)
  double* rs = hv + EROWS;              // [240] residual
  double* ri = rs + EROWS;              // [240] 1 / R
  double* sd = ri + EROWS;              // [240] diag S
  double* xs = sd + EROWS;              // [80]
  double* gv = xs + 80;                 // [32]
  double* tv = gv + 32;                 // [32]
  double* scr = tv + 32;                // FK frames, then the 25 x 27 work matrices
  Cam* cm = reinterpret_cast<Cam*>(scr + 6 * EP * SLD);
  FkLite* fr = reinterpret_cast<FkLite*>(scr);
  double* Mm = scr;                     // M (25 x 26: column 25 = g)
  double* Lc = Mm + EP * SLD;
  double* N1 = Lc + EP * SLD;           // Lc^T M, then Z = Lb^-1 [N1 | u]
  double* Bm = N1 + EP * SLD;           // B -> Lb
  double* Am = Bm + EP * SLD;
  double* V = Hq;                       // [75][27]
  double* Pt = Hq + ES * HLD;           // [25][76]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int seq = blockIdx.x, N = K.n_frames, C = K.n_cams, rows = C * 2 * NL;
  const double* det = det_all + (size_t)seq * N * C * NL * 3;
  double* x_pred = x_pred_all + (size_t)seq * N * ES;
  double* x_est = x_est_all + (size_t)seq * N * ES;
  double* P_pred = P_pred_all + (size_t)seq * N * ES * ES;
  double* P_est = P_est_all + (size_t)seq * N * ES * ES;
  const double sT = K.sT, a1 = sT, a2 = sT * sT / 2;

  for (int e = tid; e < C * ACINO_CAM_STRIDE; e += 256) reinterpret_cast<double*>(cm)[e] = cams[e];
  for (int e = tid; e < ES * PLD; e += 256) P[e] = 0.0;
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
    for (int e = tid; e < EP * EP; e += 256) {
      const int i = e / EP, j = e % EP;
      double p[3][3], r[3][3];
#pragma unroll
      for (int bi = 0; bi < 3; ++bi)
#pragma unroll
        for (int bj = 0; bj < 3; ++bj) p[bi][bj] = P[(bi * EP + i) * PLD + bj * EP + j];
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
        P[(bi * EP + i) * PLD + j] = r[bi][0] + a1 * r[bi][1] + a2 * r[bi][2] + qc[bi][0] * qd;
        P[(bi * EP + i) * PLD + EP + j] = r[bi][1] + a1 * r[bi][2] + qc[bi][1] * qd;
        P[(bi * EP + i) * PLD + 2 * EP + j] = r[bi][2] + qc[bi][2] * qd;
      }
    }
    __syncthreads();
    if (tid < ES) x_pred[(size_t)f * ES + tid] = xs[tid];
    for (int e = tid; e < ES * ES; e += 256) P_pred[(size_t)f * ES * ES + e] = P[(e / ES) * PLD + e % ES];

    // ---- measurement model: 26 pose variants (base + eps on each parameter) ----
    for (int task = tid; task < NVAR * 22; task += 256) {
      const int v = task / 22, a = task % 22;               // active angle a + 3
      int p = 0;
#pragma unroll
      for (int q = 3; q < EP; ++q) p = (c_ekf2act[q] == a + 3) ? q : p;
      const double ang = xs[p] + ((v - 1 == p) ? K.eps : 0.0);
      double s, c;
      sincos(ang, &s, &c);
      fr[v].sc[a][0] = s;
      fr[v].sc[a][1] = c;
    }
    for (int task = tid; task < NVAR * 3; task += 256) {
      const int v = task / 3, c = task % 3;
      fr[v].pos[20][c] = xs[c] + ((v - 1 == c) ? K.eps : 0.0);
    }
    __syncthreads();
    if (tid < NVAR * 3) fk_columns(fr[tid / 3], tid % 3);
    __syncthreads();
    for (int task = tid; task < NVAR * C * NL; task += 256) {
      const int v = task / (C * NL), rem = task % (C * NL), c = rem / NL, l = rem % NL;
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
    for (int e = tid; e < rows * EP; e += 256) {
      const int row = e / EP, p = e % EP;
      Hq[row * HLD + p] = (Hq[row * HLD + p] - hv[row]) / K.eps;       // (:643)
    }
    if (tid < rows) {
      const int c = tid / (2 * NL), l = (tid % (2 * NL)) / 2, d = tid & 1;
      const double* dd = det + (((size_t)f * C + c) * NL + l) * 3;
      rs[tid] = dd[d] - hv[tid];
      const double sdv = dd[2] < K.dlc_thresh ? K.max_pixel_err : 25.0;     // (:805-809: the 5**2 is squared again)
      ri[tid] = 1.0 / (sdv * sdv);
    }
    __syncthreads();
    // diag S = Hq Pq Hq^T + R, row-wise
    if (tid < rows) {
      double h[EP];
#pragma unroll
      for (int a = 0; a < EP; ++a) h[a] = Hq[tid * HLD + a];
      double s = 0.0;
      for (int a = 0; a < EP; ++a) {
        double in = 0.0;
#pragma unroll
        for (int b = 0; b < EP; ++b) in += P[a * PLD + b] * h[b];
        s += h[a] * in;
      }
      sd[tid] = s + 1.0 / ri[tid];
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
      for (int s = 0; s < rows / 4; s += 2) {             // rows / 4 = 10 C is even; operands of two steps first
        const int k0 = 4 * s + lk, k1 = k0 + 4;
        const double a0 = pa[k0 * HLD], a1 = pa[k1 * HLD];
        const double w0 = ri[k0], w1 = ri[k1];
        const double b0 = pb[k0 * sb], b1 = pb[k1 * sb];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_on ? a0 : 0.0, (b_h || b_r) ? w0 * b0 : 0.0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_on ? a1 : 0.0, (b_h || b_r) ? w1 * b1 : 0.0, acc, 0, 0, 0);
      }
      __syncthreads();      // every wave is done with the FK frames' memory (scr is reused from here on)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
        if (row < EP && col <= EP) Mm[row * SLD + col] = acc[r];
      }
    }
    for (int e = tid; e < EP * EP; e += 256) Lc[(e / EP) * SLD + e % EP] = P[(e / EP) * PLD + e % EP];
    __syncthreads();
    if (wave == 0 && !wave_chol25<0>(Lc, nullptr, lane) && !bad) { bad = true; bad_code = 2 * f + 1; }
    __syncthreads();
    // N1 = Lc^T M ;  u = Lc^T g in column 25
    for (int e = tid; e < EP * (EP + 1); e += 256) {
      const int a = e / (EP + 1), b = e % (EP + 1);
      double s = 0.0;
      for (int k = a; k < EP; ++k) s += Lc[k * SLD + a] * Mm[k * SLD + b];
      N1[a * SLD + b] = s;
    }
    __syncthreads();
    for (int e = tid; e < EP * EP; e += 256) {   // B = I + N1 Lc
      const int a = e / EP, b = e % EP;
      double s = a == b ? 1.0 : 0.0;
      for (int k = b; k < EP; ++k) s += N1[a * SLD + k] * Lc[k * SLD + b];
      Bm[a * SLD + b] = s;
    }
    __syncthreads();
    // B = Lb Lb^T and Z = Lb^-1 [N1 | u] in one wave's registers
    if (wave == 0 && !wave_chol25<EP + 1>(Bm, N1, lane) && !bad) { bad = true; bad_code = 2 * f + 2; }
    __syncthreads();
    // A = M - Z^T Z ;  w = g - Z^T zu
    for (int e = tid; e < EP * (EP + 1); e += 256) {
      const int a = e / (EP + 1), b = e % (EP + 1);
      double s = Mm[a * SLD + b];
      for (int k = 0; k < EP; ++k) s -= N1[k * SLD + a] * N1[k * SLD + b];
      if (b < EP) Am[a * SLD + b] = s;
      else gv[a] = s;
    }
    __syncthreads();
    // state correction x += P[:, :25] w ; V = P[:, :25] A ; Ptop = P[:25, :]
    if (tid < ES) {
      double s = 0.0;
      for (int a = 0; a < EP; ++a) s += P[tid * PLD + a] * gv[a];
      xs[tid] += s;
    }
    for (int e = tid; e < ES * EP; e += 256) {
      const int r = e / EP, b = e % EP;
      double s = 0.0;
      for (int a = 0; a < EP; ++a) s += P[r * PLD + a] * Am[a * SLD + b];
      V[r * HLD + b] = s;
    }
    for (int e = tid; e < EP * ES; e += 256) Pt[(e / ES) * PLD + e % ES] = P[(e / ES) * PLD + e % ES];
    __syncthreads();
    // P -= V Ptop on the matrix cores: 5 x 5 tiles of the 75 x 75 matrix (padded to 80), K = 25 padded to 28
    for (int t = wave; t < 25; t += 4) {
      const int ti = t / 5, tj = t % 5;
      const int rowA = 16 * ti + li, colB = 16 * tj + li;
      d4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r;
        acc[r] = (row < ES && colB < ES) ? P[row * PLD + colB] : 0.0;
      }
      double av[7], bv[7];
#pragma unroll
      for (int s7 = 0; s7 < 7; ++s7) {
        const int kk = 4 * s7 + lk;
        av[s7] = (rowA < ES && kk < EP) ? V[rowA * HLD + kk] : 0.0;
        bv[s7] = (colB < ES && kk < EP) ? Pt[kk * PLD + colB] : 0.0;
      }
#pragma unroll
      for (int s7 = 0; s7 < 7; ++s7) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[s7], bv[s7], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r;
        if (row < ES && colB < ES) P[row * PLD + colB] = acc[r];
      }
    }
    __syncthreads();
    if (tid < ES) x_est[(size_t)f * ES + tid] = xs[tid];
    for (int e = tid; e < ES * ES; e += 256) P_est[(size_t)f * ES * ES + e] = P[(e / ES) * PLD + e % ES];
    __syncthreads();
  }
  // outlier count: one pair per thread per frame
  for (int off = 32; off > 0; off >>= 1) n_out += __shfl_down(n_out, off, 64);
  if (lane == 0 && n_out) atomicAdd(&outliers[seq], n_out);
  if (bad && tid == 0) atomicCAS(numeric_err, 0, bad_code);   // first failure: 2 f + 1 (P) or 2 f + 2 (B)
}

// A_i = P_est[i] F^T inv(P_pred[i+1]) for i = 1 .. N-2 (one workgroup per (sequence, frame)); A_i^T = inv(P_pred[i+1]) (F P_est[i]^T).
__global__ void __launch_bounds__(256)
k_rts_gain(EkfK K, const double* __restrict__ P_pred_all, const double* __restrict__ P_est_all, double* __restrict__ A_all,
           int* __restrict__ numeric_err) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Lp = sm;                 // [75][76]
  double* G = Lp + ES * PLD;       // [75][76]
  const int tid = threadIdx.x;
  const int N = K.n_frames, per = N - 2;
  const int seq = blockIdx.x / per, i = 1 + blockIdx.x % per;
  const double* Pp = P_pred_all + ((size_t)seq * N + i + 1) * ES * ES;
  const double* Pe = P_est_all + ((size_t)seq * N + i) * ES * ES;
  double* A = A_all + ((size_t)seq * N + i) * ES * ES;
  const double a1 = K.sT, a2 = K.sT * K.sT / 2;
  for (int e = tid; e < ES * ES; e += 256) {
    const int r = e / ES, c = e % ES;
    Lp[r * PLD + c] = Pp[e];
    // G[r][c] = (F P_est^T)[r][c] = sum_k F[r][k] P_est[c][k]
    double g = Pe[c * ES + r];
    if (r < 2 * EP) g += a1 * Pe[c * ES + r + EP];
    if (r < EP) g += a2 * Pe[c * ES + r + 2 * EP];
    G[r * PLD + c] = g;
  }
  __syncthreads();
  // X = inv(P_pred) G by Gauss-Jordan elimination with partial pivoting on [P_pred | G] (the reference uses
  // np.linalg.inv, :840: LU with pivoting - no positive-definiteness requirement, which (I - K H) P does not keep
  // to round-off over thousands of frames)
  __shared__ int s_piv;
  const int lane = tid & 63;
  for (int k = 0; k < ES; ++k) {
    if (tid < 64) {
      double best = -1.0;
      int bi = k;
      for (int r = k + lane; r < ES; r += 64) {
        const double v = fabs(Lp[r * PLD + k]);
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
      const double t = M2[k * PLD + c];
      M2[k * PLD + c] = M2[pr * PLD + c];
      M2[pr * PLD + c] = t;
    }
    __syncthreads();
    const double d = Lp[k * PLD + k];
    __syncthreads();
    const double inv = 1.0 / (d != 0.0 ? d : 1.0);
    if (tid < 2 * ES) {
      double* M2 = tid < ES ? Lp : G;
      M2[k * PLD + tid % ES] *= inv;
    }
    __syncthreads();
    const int nc = (ES - k - 1) + ES;          // columns k+1.. of P_pred, all of G
    for (int e = tid; e < ES * nc; e += 256) {
      const int r = e / nc, cc = e % nc;
      if (r == k) continue;
      const double fct = Lp[r * PLD + k];
      if (cc < ES - k - 1) Lp[r * PLD + k + 1 + cc] -= fct * Lp[k * PLD + k + 1 + cc];
      else G[r * PLD + cc - (ES - k - 1)] -= fct * G[k * PLD + cc - (ES - k - 1)];
    }
    __syncthreads();
  }
  for (int e = tid; e < ES * ES; e += 256) A[e] = G[(e % ES) * PLD + e / ES];     // A = X^T
}

// smooth[i] = x_est[i] + A_i (smooth[i+1] - x_pred[i+1]), i = N-2 .. 1; frames 0 and N-1 keep the filtered state (:836-839).
__global__ void __launch_bounds__(256)
k_rts_recurse(EkfK K, const double* __restrict__ x_pred_all, const double* __restrict__ x_est_all,
              const double* __restrict__ A_all, double* __restrict__ smooth_all) {
  __shared__ double v[ES], part[3][ES];
  const int tid = threadIdx.x, N = K.n_frames, seq = blockIdx.x;
  const double* x_pred = x_pred_all + (size_t)seq * N * ES;
  const double* x_est = x_est_all + (size_t)seq * N * ES;
  const double* A = A_all + (size_t)seq * N * ES * ES;
  double* smooth = smooth_all + (size_t)seq * N * ES;
  const int r = tid % ES, p = tid / ES;      // 3 threads per row, 25 columns each
  if (tid < ES) {
    smooth[tid] = x_est[tid];
    if (N > 1) smooth[(size_t)(N - 1) * ES + tid] = x_est[(size_t)(N - 1) * ES + tid];
  }
  if (N < 3) return;
  double cur = (tid < ES) ? x_est[(size_t)(N - 1) * ES + tid] : 0.0;       // smooth[i+1][tid]
  double a[EP];
  if (tid < 3 * ES) {
#pragma unroll
    for (int c = 0; c < EP; ++c) a[c] = A[(size_t)(N - 2) * ES * ES + r * ES + p * EP + c];
  }
  for (int i = N - 2; i >= 1; --i) {
    if (tid < ES) v[tid] = cur - x_pred[(size_t)(i + 1) * ES + tid];
    __syncthreads();
    double s = 0.0;
    if (tid < 3 * ES) {
#pragma unroll
      for (int c = 0; c < EP; ++c) s += a[c] * v[p * EP + c];
      part[p][r] = s;
      if (i > 1) {   // prefetch the next gain rows while the partial sums are combined
#pragma unroll
        for (int c = 0; c < EP; ++c) a[c] = A[(size_t)(i - 1) * ES * ES + r * ES + p * EP + c];
      }
    }
    __syncthreads();
    if (tid < ES) {
      cur = x_est[(size_t)i * ES + tid] + ((part[0][tid] + part[1][tid]) + part[2][tid]);
      smooth[(size_t)i * ES + tid] = cur;
    }
  }
}

static size_t a256(size_t v) { return (v + 255) / 256 * 256; }
constexpr size_t kEkfLds = (size_t)(ES * PLD + EROWS * HLD + 4 * EROWS + 80 + 32 + 32 + 6 * EP * SLD) * sizeof(double) +
                           EKF_MAXC * sizeof(Cam);
static_assert(sizeof(FkLite) * NVAR <= 6 * EP * SLD * sizeof(double), "FK frames must fit the scratch region");
static_assert((ES * HLD + EP * PLD) <= EROWS * HLD, "V and Ptop alias the Jacobian storage");
static_assert(kEkfLds <= 160 * 1024, "EKF workgroup state must fit the 160 KB LDS");

}  // namespace acino

using namespace acino;

extern "C" {

size_t acino_sizeof_ekf_params(void) { return sizeof(acino_ekf_params); }

size_t acino_ekf_workspace_bytes(int64_t n_frames, int n_seq) {
  if (n_frames < 1 || n_seq < 1) return 0;
  const size_t N = (size_t)n_frames * n_seq;
  return 3 * a256(N * ES * ES * 8) + 2 * a256(N * ES * 8) + 1024;
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
  double* P_pred = (double*)w;  w += a256(N * ES * ES * 8);
  double* P_est = (double*)w;   w += a256(N * ES * ES * 8);
  double* A = (double*)w;       w += a256(N * ES * ES * 8);
  double* x_pred = (double*)w;  w += a256(N * ES * 8);
  int* nerr = (int*)w;
  EkfK K;
  K.n_frames = (int)prm->n_frames;
  K.n_cams = prm->n_cams;
  K.sT = 1.0 / prm->fps;
  K.dlc_thresh = prm->dlc_thresh;
  K.max_pixel_err = prm->cam_width;
  K.eps = 1e-3;
  static bool attr_done = false;
  if (!attr_done) {
    ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ekf_forward),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEkfLds));
    ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rts_gain),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * ES * PLD * 8)));
    attr_done = true;
  }
  ACINO_HIP_CHECK(hipMemsetAsync(d_outliers, 0, sizeof(int32_t) * prm->n_seq, s));
  ACINO_HIP_CHECK(hipMemsetAsync(nerr, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_ekf_forward, dim3(prm->n_seq), dim3(256), kEkfLds, s, K, d_det, d_cams24, d_states0, x_pred, d_est,
                     P_pred, P_est, d_outliers, nerr);
  ACINO_LAUNCH_CHECK();
  if (prm->n_frames >= 3) {
    hipLaunchKernelGGL(k_rts_gain, dim3((unsigned)(prm->n_seq * (prm->n_frames - 2))), dim3(256), 2 * ES * PLD * 8, s, K,
                       P_pred, P_est, A, nerr);
    ACINO_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_rts_recurse, dim3(prm->n_seq), dim3(256), 0, s, K, x_pred, d_est, A, d_smooth);
  ACINO_LAUNCH_CHECK();
  int h_err = 0;
  ACINO_HIP_CHECK(hipMemcpyAsync(&h_err, nerr, sizeof(int), hipMemcpyDeviceToHost, s));
  ACINO_HIP_CHECK(hipStreamSynchronize(s));
  if (h_err) {
    if (h_err > 0)
      set_error("EKF: %s lost positive definiteness at frame %d", (h_err & 1) ? "the pose covariance" : "I + Lc^T M Lc",
                (h_err - 1) / 2);
    else
      set_error("EKF smoother: the predicted covariance of frame %d is not positive definite", -h_err);
    return ACINO_ERR_NUMERIC;
  }
  return ACINO_OK;
}

}  // extern "C"
