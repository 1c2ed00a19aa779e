// Block cyclic reduction (BCR) for the block-tridiagonal Gauss-Newton system of the FTE solve.
//
// Unknowns are grouped in super-blocks ("nodes") of 3 frames x 25 states = 75, padded to 80 = 5 tiles
// of 16 with identity rows, so every block operation is a 5x5 grid of 16x16 fp64 tiles executed on
// the matrix cores (v_mfma_f64_16x16x4_f64).  One level = two launches:
//   elim   (one workgroup per eliminated node i with neighbours l, r):
//            D_i = L L^T (blocked Cholesky in LDS), U = L^-T by blocked inversion on the matrix cores,
//            W_l = U^T A_il, W_r = U^T A_ir (plain tile GEMMs), y = U^T b_i         -> HBM
//   update (two workgroups per remaining node j):
//            D_j -= W_r(i-)^T W_r(i-) + W_l(i+)^T W_l(i+),  b_j -= W^T y,
//            new coupling block(j', j) = -W_r(i+)^T W_l(i+)
// and back-substitution x_i = U (y - W_l x_l - W_r x_r) (three mat-vecs) walks the levels in reverse.
// Level-0 couplings are the constant third-difference blocks and are generated in LDS, never stored.
// LDS: three 80x81 fp64 matrices (155.5 KB of the 160 KB) - leading dimension 81 makes both the
// row-pattern and the column-pattern MFMA operand reads bank-conflict free.
#include "bcr.hpp"

namespace acino {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int LD = 81;
constexpr int MAT = BS * LD;
constexpr int NT = 5;

__device__ __forceinline__ double readlane_d(double x, int lane) {
  long long b = __builtin_bit_cast(long long, x);
  int lo = __builtin_amdgcn_readlane((int)b, lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// One wave: Cholesky of the 16x16 tile T (LDS, leading dim LD) in registers (lane = row, cross-lane
// broadcast by v_readlane), then the inverse of the factor (lane = column).  The tile is OVERWRITTEN by
// U_kk = (L_kk^-1)^T (upper triangular): the factor L_kk itself is not needed once its inverse exists.
__device__ void chol16_inv(double* T, int lane, int* err) {
  const int r = lane & 15;
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = (c <= r) ? T[r * LD + c] : 0.0;
  bool bad = false;
  double dinv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double ajj = readlane_d(a[j], j);
    if (!(ajj > 0.0)) {
      bad = true;
      ajj = 1.0;
    }
    const double inv = rsqrt(ajj);        // one v_rsq_f64 + refinement instead of sqrt + divide
    dinv[j] = inv;
    a[j] = (r == j) ? ajj * inv : a[j] * inv;
#pragma unroll
    for (int c = j + 1; c < 16; ++c) {
      double lc = readlane_d(a[j], c);
      a[c] = (r >= c) ? a[c] - a[j] * lc : a[c];
    }
  }
  double x[16];   // lane r holds column r of L_kk^-1
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < rr; ++k) {
      double lrk = readlane_d(a[k], rr);
      acc += lrk * x[k];
    }
    x[rr] = (rr == r) ? dinv[rr] : ((rr > r) ? -acc * dinv[rr] : 0.0);
  }
  if (lane < 16) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) T[r * LD + rr] = x[rr];     // U_kk[r][rr] = Linv_kk[rr][r]
    if (bad && err) atomicExch(err, 1);
  }
}

// Blocked Cholesky of the 80x80 matrix in LDS.  On exit: strictly-lower tiles hold L(ib,jb), diagonal
// tiles hold U_kk = (L_kk^-1)^T.  All 256 threads.
__device__ void chol80(double* Lm, int tid, int* err) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  for (int kb = 0; kb < NT; ++kb) {
    double* Ukk = Lm + (kb * 16) * LD + kb * 16;
    if (wave == 0) chol16_inv(Ukk, lane, err);
    __syncthreads();
    for (int ib = kb + 1 + wave; ib < NT; ib += 4) {  // panel: L(ib,kb) = A(ib,kb) * Linv_kk^T = A(ib,kb) * U_kk
      double* A = Lm + (ib * 16) * LD + kb * 16;
      double av[4], bv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        av[s] = A[li * LD + 4 * s + lk];
        bv[s] = Ukk[(4 * s + lk) * LD + li];
      }
      d4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma(av[s], bv[s], acc);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) A[(lk + 4 * rr) * LD + li] = acc[rr];
    }
    __syncthreads();
    int cnt = 0;
    for (int ib = kb + 1; ib < NT; ++ib)
      for (int jb = kb + 1; jb <= ib; ++jb) {
        if ((cnt++ & 3) != wave) continue;
        double* Cc = Lm + (ib * 16) * LD + jb * 16;
        const double* A = Lm + (ib * 16) * LD + kb * 16;
        const double* B = Lm + (jb * 16) * LD + kb * 16;
        d4 acc;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[rr] = Cc[(lk + 4 * rr) * LD + li];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(-A[li * LD + 4 * s + lk], B[li * LD + 4 * s + lk], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = acc[rr];
      }
    __syncthreads();
  }
}

// Blocked inversion of the Cholesky factor: fills the strictly-upper tiles with U = (L^-1)^T, i.e. tile
// (jb, ib) = X(ib,jb)^T where X = L^-1, X(ib,jb) = -X(ib,ib) * sum_{k=jb}^{ib-1} L(ib,k) X(k,jb).
// Block column jb is one wave's sequential chain (no workgroup barrier inside); 4 waves = columns 0..3.
__device__ void linv80(double* Lm, int tid) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int jb = wave;
  if (jb >= NT - 1) return;
  for (int ib = jb + 1; ib < NT; ++ib) {
    d4 t = {0, 0, 0, 0};
    for (int k = jb; k < ib; ++k) {
      const double* A = Lm + (ib * 16) * LD + k * 16;         // L(ib,k)[i][kk]
      // X(k,jb)[kk][j] = U[jb16+j][k16+kk]  (diagonal tile k == jb included: U_jj[j][kk] = X_jj[kk][j])
      const double* B = Lm + (jb * 16) * LD + k * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma(A[li * LD + 4 * s + lk], B[li * LD + 4 * s + lk], t);
    }
    const double* Uii = Lm + (ib * 16) * LD + ib * 16;         // X(ib,ib)[i][kk] = U_ii[kk][i]
    d4 x = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) x = mfma(-Uii[(4 * s + lk) * LD + li], t[s], x);
    double* Ut = Lm + (jb * 16) * LD + ib * 16;                 // tile (jb, ib) <- X(ib,jb)^T
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Ut[li * LD + lk + 4 * rr] = x[rr];
  }
}

// W <- L^-1 W = U^T W for the column strip starting at column cc (one wave, in place, descending row tiles).
__device__ __forceinline__ void linv_gemm_strip(const double* Lm, double* W, int cc, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  for (int ib = NT - 1; ib >= 0; --ib) {
    d4 acc = {0, 0, 0, 0};
    for (int k = 0; k <= ib; ++k) {
      const double* Uk = Lm + (k * 16) * LD + ib * 16;          // X(ib,k)[i][kk] = U[k16+kk][ib16+i]
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = mfma(Uk[(4 * s + lk) * LD + li], W[(k * 16 + 4 * s + lk) * LD + cc + li], acc);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) W[(ib * 16 + lk + 4 * rr) * LD + cc + li] = acc[rr];
  }
}

// 80x80 fp64 matrix HBM <-> LDS with all loads of a thread in flight before the first use (13 x 16 B).
template <bool TRANSPOSE>
__device__ __forceinline__ void load_mat_any(double* dst, const double* __restrict__ src, int tid) {
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2 v[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int idx = tid + 256 * k;
    if (idx < BS * BS / 2) v[k] = s2[idx];
  }
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int idx = tid + 256 * k;
    if (idx < BS * BS / 2) {
      const int e = 2 * idx, r = e / BS, c = e % BS;
      if (TRANSPOSE) {
        dst[c * LD + r] = v[k].x;
        dst[(c + 1) * LD + r] = v[k].y;
      } else {
        dst[r * LD + c] = v[k].x;
        dst[r * LD + c + 1] = v[k].y;
      }
    }
  }
}
__device__ __forceinline__ void load_mat(double* dst, const double* __restrict__ src, int tid) {
  load_mat_any<false>(dst, src, tid);
}
__device__ __forceinline__ void load_mat_t(double* dst, const double* __restrict__ src, int tid) {
  load_mat_any<true>(dst, src, tid);
}
__device__ __forceinline__ void store_mat(double* __restrict__ dst, const double* src, int tid) {
  double2* d2 = reinterpret_cast<double2*>(dst);
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int idx = tid + 256 * k;
    if (idx < BS * BS / 2) {
      const int e = 2 * idx, r = e / BS, c = e % BS;
      d2[idx] = make_double2(src[r * LD + c], src[r * LD + c + 1]);
    }
  }
}

// Analytic level-0 coupling between node `i` (rows) and its chain neighbour (cols): third-difference blocks.
__device__ void gen_coupling(double* W, const FteConst& K, int node_i, bool left, int tid) {
  for (int e = tid; e < BS * LD; e += 256) W[e] = 0.0;
  __syncthreads();
  const int64_t f_i = K.n_offset + 3 * (int64_t)(node_i - K.pin_left);
  for (int e = tid; e < 9 * NP; e += 256) {
    int p = e % NP, pair = e / NP, ii = pair / 3, jj = pair % 3;
    if (ii > jj) continue;
    int k = 3 + ii - jj;
    if (left) {   // rows (ii,p) of node i, cols (jj,p) of node i-1 ; column frame is the earlier one
      double v = 2.0 * K.q_w[p] * band_coef(f_i - 3 + jj, k, K.n_global);
      W[(ii * NP + p) * LD + jj * NP + p] = v;
    } else {      // rows (jj,p) of node i, cols (ii,p) of node i+1 ; row frame is the earlier one
      double v = 2.0 * K.q_w[p] * band_coef(f_i + jj, k, K.n_global);
      W[(jj * NP + p) * LD + ii * NP + p] = v;
    }
  }
}

// Eliminate node i: D_i = L L^T, U = L^-T, W_l = U^T A_il, W_r = U^T A_ir, y = U^T b_i.  Stores U (in the
// D slot), W_l, W_r (in the coupling slot) and y.
__global__ void __launch_bounds__(256)
k_bcr_elim(BcrChain ch, const int* __restrict__ elim, const FteConst* __restrict__ cst, int* numeric_err,
           const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Lm = reinterpret_cast<double*>(smem_raw);
  double* WL = Lm + MAT;
  double* WR = WL + MAT;
  double* yv = WR + MAT;       // [80] rhs
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = elim[3 * blockIdx.x], l = elim[3 * blockIdx.x + 1], r = elim[3 * blockIdx.x + 2];
  const size_t MB = (size_t)BS * BS;
#define ACINO_STAMP(k) do { if (ch.dbg && blockIdx.x == 0 && tid == 0) ch.dbg[k] = (long long)wall_clock64(); } while (0)
  ACINO_STAMP(0);
  load_mat(Lm, ch.D + i * MB, tid);
  if (tid < BS) yv[tid] = ch.b[(size_t)i * BS + tid];
  if (l >= 0) {
    if (ch.implicit_couplings && l == i - 1) gen_coupling(WL, *cst, i, true, tid);
    else load_mat(WL, ch.Cpl + l * MB, tid);           // block(i, l): rows i, cols l
  }
  if (r >= 0) {
    if (ch.implicit_couplings && r == i + 1) gen_coupling(WR, *cst, i, false, tid);
    else load_mat_t(WR, ch.Cpl + i * MB, tid);         // block(r, i)^T: rows i, cols r
  }
  __syncthreads();
  ACINO_STAMP(1);
  chol80(Lm, tid, numeric_err);
  ACINO_STAMP(2);
  linv80(Lm, tid);
  __syncthreads();
  ACINO_STAMP(3);
  for (int ct = wave; ct < 10; ct += 4) {               // W_l, W_r: ten column strips over four waves
    if (ct < 5) {
      if (l >= 0) linv_gemm_strip(Lm, WL, ct * 16, lane);
    } else {
      if (r >= 0) linv_gemm_strip(Lm, WR, (ct - 5) * 16, lane);
    }
  }
  double yy = 0.0;
  if (tid < BS)
    for (int c = 0; c <= tid; ++c) yy += Lm[c * LD + tid] * yv[c];   // y = U^T b
  __syncthreads();
  ACINO_STAMP(4);
  store_mat(ch.D + i * MB, Lm, tid);
  if (l >= 0) store_mat(ch.Wl + i * MB, WL, tid);
  if (r >= 0) store_mat(ch.Cpl + i * MB, WR, tid);
  if (tid < BS) ch.b[(size_t)i * BS + tid] = yy;
  __syncthreads();
  ACINO_STAMP(5);
}

// lower-triangular tile enumeration t -> (ib, jb)
__constant__ int8_t c_tri_i[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4};
__constant__ int8_t c_tri_j[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};

// role 0: D_j / b_j update; role 1: new coupling block(jn, j)
__global__ void __launch_bounds__(256)
k_bcr_update(BcrChain ch, const int* __restrict__ remain, const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* WA = reinterpret_cast<double*>(smem_raw);
  double* WB = WA + MAT;
  double* ya = WB + MAT;
  double* yb = ya + BS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int ent = blockIdx.x >> 1, role = blockIdx.x & 1;
  const int j = remain[4 * ent], im = remain[4 * ent + 1], ip = remain[4 * ent + 2], jn = remain[4 * ent + 3];
  const size_t MB = (size_t)BS * BS;
  if (role == 0) {
    if (im < 0 && ip < 0) return;
    if (im >= 0) {
      load_mat(WA, ch.Cpl + im * MB, tid);   // W_r of the eliminated left neighbour (cols = j)
      if (tid < BS) ya[tid] = ch.b[(size_t)im * BS + tid];
    }
    if (ip >= 0) {
      load_mat(WB, ch.Wl + ip * MB, tid);    // W_l of the eliminated right neighbour (cols = j)
      if (tid < BS) yb[tid] = ch.b[(size_t)ip * BS + tid];
    }
    double* Dj = ch.D + j * MB;
    // this wave's (<= 4) output tiles of D_j (C layout) are fetched while the W matrices land in LDS
    d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      if (t < 15) {
        const int ib = c_tri_i[t], jb = c_tri_j[t];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[q][rr] = Dj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      if (t < 15) {
        const int ib = c_tri_i[t], jb = c_tri_j[t];
        d4 a = acc[q];
        if (im >= 0)
          for (int s = 0; s < BS / 4; ++s)
            a = mfma(-WA[(4 * s + lk) * LD + ib * 16 + li], WA[(4 * s + lk) * LD + jb * 16 + li], a);
        if (ip >= 0)
          for (int s = 0; s < BS / 4; ++s)
            a = mfma(-WB[(4 * s + lk) * LD + ib * 16 + li], WB[(4 * s + lk) * LD + jb * 16 + li], a);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          Dj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = a[rr];
          if (ib != jb) Dj[(jb * 16 + li) * BS + ib * 16 + lk + 4 * rr] = a[rr];
        }
      }
    }
    if (tid < BS) {
      double s = ch.b[(size_t)j * BS + tid];
      if (im >= 0)
        for (int k = 0; k < BS; ++k) s -= WA[k * LD + tid] * ya[k];
      if (ip >= 0)
        for (int k = 0; k < BS; ++k) s -= WB[k * LD + tid] * yb[k];
      ch.b[(size_t)j * BS + tid] = s;
    }
  } else {
    if (ip < 0 || jn < 0) return;
    load_mat(WA, ch.Cpl + ip * MB, tid);     // W_r(ip): cols = jn
    load_mat(WB, ch.Wl + ip * MB, tid);      // W_l(ip): cols = j
    __syncthreads();
    double* Cj = ch.Cpl + j * MB;            // block(jn, j): rows jn, cols j
    for (int t = wave; t < NT * NT; t += 4) {
      const int ib = t / NT, jb = t % NT;
      d4 acc = {0, 0, 0, 0};
      for (int s = 0; s < BS / 4; ++s)
        acc = mfma(-WA[(4 * s + lk) * LD + ib * 16 + li], WB[(4 * s + lk) * LD + jb * 16 + li], acc);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Cj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = acc[rr];
    }
  }
}

// x_i = U (y_i - W_l x_l - W_r x_r),  U = L^-T
__global__ void __launch_bounds__(256)
k_bcr_backsub(BcrChain ch, const int* __restrict__ elim, const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Um = reinterpret_cast<double*>(smem_raw);
  double* WL = Um + MAT;
  double* WR = WL + MAT;
  double* xl = WR + MAT;
  double* xr = xl + BS;
  double* tv = xr + BS;
  const int tid = threadIdx.x;
  const int i = elim[3 * blockIdx.x], l = elim[3 * blockIdx.x + 1], r = elim[3 * blockIdx.x + 2];
  const size_t MB = (size_t)BS * BS;
  load_mat(Um, ch.D + i * MB, tid);
  if (l >= 0) load_mat(WL, ch.Wl + i * MB, tid);
  if (r >= 0) load_mat(WR, ch.Cpl + i * MB, tid);
  if (tid < BS) {
    xl[tid] = l >= 0 ? ch.b[(size_t)l * BS + tid] : 0.0;
    xr[tid] = r >= 0 ? ch.b[(size_t)r * BS + tid] : 0.0;
  }
  __syncthreads();
  if (tid < BS) {                          // t = y - W_l x_l - W_r x_r  (row tid; stride-81 rows: conflict free)
    double s0 = ch.b[(size_t)i * BS + tid], s1 = 0.0;
    if (l >= 0)
      for (int c = 0; c < BS; ++c) s0 -= WL[tid * LD + c] * xl[c];
    if (r >= 0)
      for (int c = 0; c < BS; ++c) s1 -= WR[tid * LD + c] * xr[c];
    tv[tid] = s0 + s1;
  }
  __syncthreads();
  if (tid < BS) {                          // x = U t (U upper triangular)
    double s = 0.0;
    for (int c = tid; c < BS; ++c) s += Um[tid * LD + c] * tv[c];
    ch.b[(size_t)i * BS + tid] = s;
  }
}

// ---- host side ------------------------------------------------------------------------------
void BcrSchedule::build(int n, bool pin_left, bool pin_right) {
  levels.clear();
  elim.clear();
  remain.clear();
  std::vector<int> act(n);
  for (int i = 0; i < n; ++i) act[i] = i;
  auto pinned = [&](int node) { return (pin_left && node == 0) || (pin_right && node == n - 1); };
  while (true) {
    const int R = (int)act.size();
    std::vector<char> pick(R, 0);
    int n_pick = 0;
    for (int p = 0; p < R; ++p)
      if (!pinned(act[p]) && (p == 0 || !pick[p - 1])) {
        pick[p] = 1;
        ++n_pick;
      }
    if (n_pick == 0) break;
    BcrLevel lv;
    lv.elim_off = (int)elim.size() / 3;
    lv.remain_off = (int)remain.size() / 4;
    lv.n_elim = n_pick;
    std::vector<int> next;
    for (int p = 0; p < R; ++p) {
      if (pick[p]) {
        elim.push_back(act[p]);
        elim.push_back(p > 0 ? act[p - 1] : -1);
        elim.push_back(p + 1 < R ? act[p + 1] : -1);
      } else {
        next.push_back(act[p]);
      }
    }
    int n_rem = 0;
    for (int p = 0; p < R; ++p) {
      if (pick[p]) continue;
      int im = (p > 0 && pick[p - 1]) ? act[p - 1] : -1;
      int ip = (p + 1 < R && pick[p + 1]) ? act[p + 1] : -1;
      int jn = (ip >= 0 && p + 2 < R) ? act[p + 2] : -1;
      if (im < 0 && ip < 0) continue;
      remain.push_back(act[p]);
      remain.push_back(im);
      remain.push_back(ip);
      remain.push_back(jn);
      ++n_rem;
    }
    lv.n_remain = n_rem;
    levels.push_back(lv);
    act.swap(next);
    if (act.empty()) break;
  }
}

static constexpr size_t kElimLds = (3 * MAT + 2 * BS) * sizeof(double);
static constexpr size_t kUpdateLds = (2 * MAT + 2 * BS) * sizeof(double);
static constexpr size_t kBacksubLds = (3 * MAT + 3 * BS) * sizeof(double);

int bcr_set_func_attributes() {
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_elim),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kElimLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_update),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpdateLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_backsub),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubLds));
  return ACINO_OK;
}

int bcr_reduce(const BcrChain& ch, const BcrSchedule& sch, const FteConst* d_c, int* d_numeric_err,
               const int* d_status, hipStream_t s, Profiler* prof) {
  for (const BcrLevel& lv : sch.levels) {
    {
      ProfSpan sp(prof, PC_ELIM, s);
      hipLaunchKernelGGL(k_bcr_elim, dim3(lv.n_elim), dim3(256), kElimLds, s, ch, ch.d_elim + 3 * lv.elim_off,
                         d_c, d_numeric_err, d_status);
    }
    ACINO_LAUNCH_CHECK();
    if (lv.n_remain > 0) {
      {
        ProfSpan sp(prof, PC_UPDATE, s);
        hipLaunchKernelGGL(k_bcr_update, dim3(2 * lv.n_remain), dim3(256), kUpdateLds, s, ch,
                           ch.d_remain + 4 * lv.remain_off, d_status);
      }
      ACINO_LAUNCH_CHECK();
    }
  }
  return ACINO_OK;
}

int bcr_backsub(const BcrChain& ch, const BcrSchedule& sch, const int* d_status, hipStream_t s, Profiler* prof) {
  for (int k = (int)sch.levels.size() - 1; k >= 0; --k) {
    const BcrLevel& lv = sch.levels[k];
    {
      ProfSpan sp(prof, PC_BACKSUB, s);
      hipLaunchKernelGGL(k_bcr_backsub, dim3(lv.n_elim), dim3(256), kBacksubLds, s, ch,
                         ch.d_elim + 3 * lv.elim_off, d_status);
    }
    ACINO_LAUNCH_CHECK();
  }
  return ACINO_OK;
}

// ---- MFMA layout self-test: C[16][16] = A[16][K] * B[K][16] ---------------------------------
__global__ void k_selftest_mfma(const double* a, const double* b, int k, double* c) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  d4 acc = {0, 0, 0, 0};
  for (int s = 0; s < k / 4; ++s) acc = mfma(a[li * k + 4 * s + lk], b[(4 * s + lk) * 16 + li], acc);
  for (int rr = 0; rr < 4; ++rr) c[(lk + 4 * rr) * 16 + li] = acc[rr];
}

}  // namespace acino

extern "C" int acino_selftest_mfma(const double* d_a, const double* d_b, int k, double* d_c, void* stream) {
  using namespace acino;
  ACINO_REQUIRE(k > 0 && k % 4 == 0, "k must be a positive multiple of 4");
  ACINO_REQUIRE(d_a && d_b && d_c, "null buffer");
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, d_a, d_b, k, d_c);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}
