// Chunked substructuring of the block-tridiagonal Gauss-Newton system (the step SURVEY.md section 7.6 describes):
// the chain of 80x80 super-blocks is cut into runs of m consecutive nodes; ONE workgroup eliminates the m - 1 interior
// nodes of a run IN ORDER with every operand resident in LDS, the last node of each run is a separator.  In sequential
// order the couplings between consecutive interior nodes stay the constant third-difference stencils E (<= 3 terms per
// entry, never stored); the only dense fill is the "spike" F_k = block(node k, left separator L):
//
//   node k (D~_k, F_k, b~_k in LDS):   D~_k = L L^T,  U = L^-T  (blocked Cholesky, dense80.hpp)
//        y = U^T b~_k            z_k = U y                       (= D~_k^-1 b~_k)
//        W = U^T F_k             T_k = U W                       (= D~_k^-1 F_k)
//        G_k = U U^T                                              (= D~_k^-1)
//      left separator:   D_L -= W^T W        b_L -= W^T y
//      next node:        D~_k+1 = D_k+1 - E^T G_k E    F_k+1 = -E^T T_k    b~_k+1 = b_k+1 - E^T z_k     (E = E_r(k))
//      (after the last interior node the "next node" is the right separator R: F becomes block(R, L))
//   back-substitution, right to left:  x_k = z_k - G_k (E x_k+1) - T_k x_L.
//
// HBM traffic per interior node: H, g in (15 KB), G_k, T_k, z_k out (102 KB) and in again for the back-substitution;
// the block cyclic reduction of the same nodes moved D, U, W_l, W_r and the coupling blocks through HBM at every level.
// The separators (one per run, n_chunks - 1) are a block-tridiagonal chain with dense couplings: bcr.hip solves it.
#include "chunk.hpp"

#include <algorithm>
#include <cstdlib>

#include "bcr_dev.hpp"

namespace acino {

void ChunkPlan::build(int nodes, int chunk_nodes) {
  n_nodes = nodes;
  m = n_chunks = n_sep = 0;
  if (chunk_nodes < 0 || nodes < 1) return;
  int mm = chunk_nodes;
  if (mm == 0) {
    // automatic: one run per CU (256) - one round of workgroups, the fewest separators - for the long chains (measured on
    // config 5's 21 334-node chain: 3.45 ms per iteration with runs of 84 against 3.90 with runs of 16 in 5 rounds), runs of
    // at least 4 nodes for the short ones
    mm = (nodes + 255) / 256;
    mm = std::max(4, std::min(mm, 512));
  }
  mm = std::max(2, mm);
  m = mm;
  n_chunks = (nodes + mm - 1) / mm;
  n_sep = n_chunks - 1;
}

// ---- tile products on LDS matrices (leading dimension LD), one 16x16 output tile per call, all operands read first ----
// (U^T F)(ib, jb) = sum_{kb <= ib} U(kb, ib)^T F(kb, jb): U upper triangular in X (strictly-lower tiles are workspace)
__device__ __forceinline__ d4 tile_ut_f(const double* X, const double* Y, int ib, int jb, int li, int lk) {
  const double* pa = X + lk * LD + ib * 16 + li;
  const double* pb = Y + lk * LD + jb * 16 + li;
  const d4 z = {0, 0, 0, 0};
  switch (ib) {
    case 0: return mma_seq<4, false>(z, pa, 4 * LD, pb, 4 * LD);
    case 1: return mma_seq<8, false>(z, pa, 4 * LD, pb, 4 * LD);
    case 2: return mma_seq<12, false>(z, pa, 4 * LD, pb, 4 * LD);
    case 3: return mma_seq<16, false>(z, pa, 4 * LD, pb, 4 * LD);
    default: return mma_seq<20, false>(z, pa, 4 * LD, pb, 4 * LD);
  }
}
// (U W)(ib, jb) = sum_{kb >= ib} U(ib, kb) W(kb, jb)
__device__ __forceinline__ d4 tile_u_w(const double* X, const double* Y, int ib, int jb, int li, int lk) {
  const double* pa = X + (ib * 16 + li) * LD + ib * 16 + lk;
  const double* pb = Y + (ib * 16 + lk) * LD + jb * 16 + li;
  const d4 z = {0, 0, 0, 0};
  switch (ib) {
    case 0: return mma_seq<20, false>(z, pa, 4, pb, 4 * LD);
    case 1: return mma_seq<16, false>(z, pa, 4, pb, 4 * LD);
    case 2: return mma_seq<12, false>(z, pa, 4, pb, 4 * LD);
    case 3: return mma_seq<8, false>(z, pa, 4, pb, 4 * LD);
    default: return mma_seq<4, false>(z, pa, 4, pb, 4 * LD);
  }
}
// (U U^T)(ib, jb), jb <= ib:  sum_{k >= 16 ib} U[ib16 + i][k] U[jb16 + j][k]
__device__ __forceinline__ d4 tile_u_ut(const double* X, int ib, int jb, int li, int lk) {
  const double* pa = X + (ib * 16 + li) * LD + ib * 16 + lk;
  const double* pb = X + (jb * 16 + li) * LD + ib * 16 + lk;
  const d4 z = {0, 0, 0, 0};
  switch (ib) {
    case 0: return mma_seq<20, false>(z, pa, 4, pb, 4);
    case 1: return mma_seq<16, false>(z, pa, 4, pb, 4);
    case 2: return mma_seq<12, false>(z, pa, 4, pb, 4);
    case 3: return mma_seq<8, false>(z, pa, 4, pb, 4);
    default: return mma_seq<4, false>(z, pa, 4, pb, 4);
  }
}
__device__ __forceinline__ void tile_store(double* M, int ib, int jb, const d4& a, int li, int lk) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) M[(ib * 16 + lk + 4 * rr) * LD + jb * 16 + li] = a[rr];
}

// ================================================================================================================
// The sweep kernel.  Per node the work splits into a SERIAL part (all four waves) and a PARALLEL part:
//   serial   : G_k = U_k U_k^T -> HBM;  D~_k+1 = D_k+1 - E^T G_k E  (the next node's H / g / x were requested before G)
//   parallel : wave 0 factors D~_k+1 ALONE (one-wave blocked Cholesky: no workgroup barrier on the pivot chain),
//              waves 1..3 do the spike algebra of node k meanwhile: W = U_k^T F_k, D_L -= W^T W, T_k = U_k W -> HBM,
//              F_k+1 = -E^T T_k, each wave on its own 16-column strips (two 3-wave LDS barriers per node for W^T W).
// The right-hand side rides along as COLUMN 79 of the spike (the left separator has 75 unknowns, columns 75..79 are free):
// column 79 of W is y = U^T b~, of T it is z = D~^-1 b~, of F_k+1 it is -E^T z (+ b_k+1 = the next node's b~), and row 79
// of W^T W is y^T W = the left separator's right-hand-side update - no mat-vec phases, no extra barriers.
// LDS: three 80 x 81 matrices (U_k | D~_k+1 -> U_k+1 | spike) + tables = 159.9 KB, one workgroup per CU.
constexpr int SW_VEC = 18 * NP + BS + 8 + 8 + 3 * NP + 64;   // cL cR | bv | red | sync | kq klo khi | debug stamps
static constexpr size_t kSweepLds = (3 * MAT + SW_VEC) * sizeof(double);

// The value of x, made opaque to the optimiser: address arithmetic derived from it cannot be hoisted out of the node loop
// (hoisted, the hundreds of per-tile LDS / HBM addresses of this kernel end up spilled to scratch and are reloaded one by
// one in front of the loads that need them).
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
template <class T>
__device__ __forceinline__ T* opaque_ptr(T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// Barrier among n waves of the workgroup that share the LDS counter cnt (the other waves are busy elsewhere and must not
// be waited for, so s_barrier cannot be used): every participant adds one and waits for the n-th arrival of this round.
__device__ __forceinline__ void sub_barrier(int* cnt, int& target, int n, int lane) {
  target += n;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Trailing phase of step KB of the blocked Cholesky by ONE wave: every tile product is C(ti, tj) -= P(ti) P(tj)^T with
// P(t) = tile (t, KB) (the panel output; P(KB) = U_kk for the tiles of U), and both operands of a product are read with the
// SAME lane pattern - so the five P tiles are read once into registers (20 doubles) and serve all 13 / 11 / 8 / 4 products
// of the step, whose addresses are compile-time constants.  The tiles the next pivot chain needs come first.
template <int KB>
struct TrailList {
  int ti[13], tj[13], n;
  constexpr TrailList() : ti{}, tj{}, n(0) {
    for (int r = KB + 2; r < NT; ++r)
      for (int c = KB + 1; c <= r; ++c) {
        ti[n] = r;
        tj[n] = c;
        ++n;
      }
    for (int r = 0; r <= KB; ++r)
      for (int c = KB + 1; c < NT; ++c) {
        ti[n] = r;
        tj[n] = c;
        ++n;
      }
  }
};
template <int KB, int PART = 0, int NPARTS = 1>
__device__ __forceinline__ void trail_step(double* Lm, int li, int lk) {
  double P[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) P[t][s] = Lm[(t * 16 + li) * LD + KB * 16 + 4 * s + lk];
  // the products of the step as a compile-time list (ti, tj): lower part first - rows KB+2 .. 4, row KB+2 feeds the next
  // look-ahead (the look-ahead tile (KB+1, KB+1) itself is the chain wave's) -, then the tiles of U (ti <= KB < tj, first
  // written at ti == KB); processed four at a time with their accumulator chains interleaved
  // (PART of NPARTS: products PART, PART + NPARTS, ... of the list - two helper waves take alternate products)
  constexpr TrailList<KB> TL{};
  constexpr int NTOT = (TL.n - PART + NPARTS - 1) / NPARTS;
  auto tile_of = [&](int q, int& ti, int& tj) {
    ti = TL.ti[PART + NPARTS * q];
    tj = TL.tj[PART + NPARTS * q];
  };
#pragma unroll
  for (int q0 = 0; q0 < NTOT; q0 += 4) {
    d4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q0 + q < NTOT) {
        int ti = 0, tj = 0;
        tile_of(q0 + q, ti, tj);
        const double* Cc = Lm + (ti * 16) * LD + tj * 16;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a[q][rr] = ti == KB ? 0.0 : Cc[(lk + 4 * rr) * LD + li];
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q0 + q < NTOT) {
          int ti = 0, tj = 0;
          tile_of(q0 + q, ti, tj);
          a[q] = mfma(-P[ti][s], P[tj][s], a[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q0 + q < NTOT) {
        int ti = 0, tj = 0;
        tile_of(q0 + q, ti, tj);
        double* Cc = Lm + (ti * 16) * LD + tj * 16;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = a[q][rr];
      }
    }
  }
}

// chol80 by a PAIR of waves: role 0 runs the five 16-pivot chains with the look-ahead update of the next diagonal tile
// (exactly wave 0 of chol80), role 1 does every other panel and trailing tile, batched four at a time.
__device__ __forceinline__ void chol80_pair(double* Lm, int role, int lane, int* err, int* cnt, int& target, int& posted,
                                            long long* dbg = nullptr) {
  const int li = lane & 15, lk = lane >> 4;
#define CH_STAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  if (role == 0) chol16_inv(Lm, lane, err);
  CH_STAMP(32 + 16 * role);
  sub_barrier(cnt, target, 2, lane);
  CH_STAMP(33 + 16 * role);
#pragma unroll 1
  for (int kb = 0; kb < NT; ++kb) {
    {  // panel: tile(t, kb) <- tile(t, kb) U_kk: role 0 the tile its look-ahead needs next, role 1 the other three
      const double* Ukk = Lm + (kb * 16) * LD + kb * 16;
      double bq[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bq[s] = Ukk[(4 * s + lk) * LD + li];
      if (role == 0) {
        double* A = Lm + (panel_tile(kb, 0) * 16) * LD + kb * 16;
        double av[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = A[li * LD + 4 * s + lk];
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(av[s], bq[s], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) A[(lk + 4 * rr) * LD + li] = acc[rr];
      } else {
        double av[3][4];
        d4 acc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const double* A = Lm + (panel_tile(kb, q + 1) * 16) * LD + kb * 16;
#pragma unroll
          for (int s = 0; s < 4; ++s) av[q][s] = A[li * LD + 4 * s + lk];
          acc[q] = d4{0, 0, 0, 0};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int q = 0; q < 3; ++q) acc[q] = mfma(av[q][s], bq[s], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          double* A = Lm + (panel_tile(kb, q + 1) * 16) * LD + kb * 16;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) A[(lk + 4 * rr) * LD + li] = acc[q][rr];
        }
      }
    }
    CH_STAMP(34 + 16 * role + 3 * kb);
    if (kb == NT - 1) {
      sub_barrier(cnt, target, 2, lane);
      break;
    }
    // role 0 goes straight on with the look-ahead (it needs only its own panel tile); role 1's trailing products also
    // read that tile: one-way flag cnt[3] (monotonic: `posted` panel tiles so far)
    ++posted;
    if (role == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(cnt + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      while (__hip_atomic_load(cnt + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < posted) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    CH_STAMP(35 + 16 * role + 3 * kb);
    if (role == 0) {                          // next diagonal tile, then its 16-pivot chain
      double* Cc = Lm + ((kb + 1) * 16) * LD + (kb + 1) * 16;
      const double* A = Lm + ((kb + 1) * 16) * LD + kb * 16;
      d4 a;
      double av[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(lk + 4 * rr) * LD + li];
#pragma unroll
      for (int s = 0; s < 4; ++s) av[s] = A[li * LD + 4 * s + lk];
#pragma unroll
      for (int s = 0; s < 4; ++s) a = mfma(-av[s], av[s], a);
      chol16_inv_acc(Cc, a, lane, err);
    } else {                                  // the other trailing tiles (U tiles included): 13 | 11 | 8 | 4
      if (kb == 0) trail_step<0>(Lm, li, lk);
      else if (kb == 1) trail_step<1>(Lm, li, lk);
      else if (kb == 2) trail_step<2>(Lm, li, lk);
      else trail_step<3>(Lm, li, lk);
    }
    CH_STAMP(36 + 16 * role + 3 * kb);
    sub_barrier(cnt, target, 2, lane);
  }
#undef CH_STAMP
}

// W(:, strips) = U^T F(:, strips), in place in Y, for NS 16-column strips of one wave: W(ib, jb) = sum_{kb <= ib}
// U(kb, ib)^T F(kb, jb).  k loop outermost: the U operand of a k-step is shared by the NS strips, the F operand by the row
// tiles ib >= kb, and the NS (5 - kb) accumulator chains of a step are independent.
template <int NS>
__device__ __forceinline__ void strips_ut_f(const double* U, double* Y, const int (&jbs)[NS], int li, int lk) {
  d4 acc[NS][NT];
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) acc[j][ib] = d4{0, 0, 0, 0};
#pragma unroll
  for (int kb = 0; kb < NT; ++kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 16 * kb + 4 * s + lk;
      double b[NS], a[NT];
#pragma unroll
      for (int j = 0; j < NS; ++j) b[j] = Y[k * LD + jbs[j] * 16 + li];
#pragma unroll
      for (int ib = kb; ib < NT; ++ib) a[ib] = U[k * LD + ib * 16 + li];
#pragma unroll
      for (int ib = kb; ib < NT; ++ib)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[j][ib] = mfma(a[ib], b[j], acc[j][ib]);
    }
  }
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, jbs[j], acc[j][ib], li, lk);
}
// T(:, strips) = U W(:, strips): T(ib, jb) = sum_{kb >= ib} U(ib, kb) W(kb, jb); accumulators out (the caller stores them
// once nobody reads W any more)
template <int NS>
__device__ __forceinline__ void strips_u_w_acc(const double* U, const double* Y, const int (&jbs)[NS], d4 (&acc)[NS][NT], int li,
                                               int lk) {
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) acc[j][ib] = d4{0, 0, 0, 0};
#pragma unroll
  for (int kb = 0; kb < NT; ++kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 16 * kb + 4 * s + lk;
      double b[NS], a[NT];
#pragma unroll
      for (int j = 0; j < NS; ++j) b[j] = Y[k * LD + jbs[j] * 16 + li];
#pragma unroll
      for (int ib = 0; ib <= kb; ++ib) a[ib] = U[(ib * 16 + li) * LD + k];
#pragma unroll
      for (int ib = 0; ib <= kb; ++ib)
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[j][ib] = mfma(a[ib], b[j], acc[j][ib]);
    }
  }
}

// One 16-column strip (tile column jb) of a triangular 80 x 80 product, five accumulator tiles, operands of k-step
// s + 2 requested before the matrix-core work of k-step s (the compiler does not pipeline the loop by itself: it waits for
// a step's operands, issues the step's products and only then requests the next operands).
//   UT = true :  W(:, jb) = U^T F(:, jb),  W(ib, jb) = sum_{kb <= ib} U(kb, ib)^T F(kb, jb)   (in place in Y by the caller)
//   UT = false:  T(:, jb) = U W(:, jb),    T(ib, jb) = sum_{kb >= ib} U(ib, kb) W(kb, jb)
template <bool UT>
__device__ __forceinline__ void strip_product(const double* U, const double* Y, int jb, d4 (&acc)[NT], int li, int lk) {
  constexpr int NSTEP = 4 * NT;
  double a[3][NT], b[3];
  const double* yb = Y + lk * LD + jb * 16 + li;
  const double* ub = UT ? U + lk * LD + li : U + li * LD + lk;
  auto fetch = [&](int buf, int step) {
    const int kb = step >> 2;
    b[buf] = yb[(4 * step) * LD];
#pragma unroll
    for (int ib = 0; ib < NT; ++ib)
      if (UT ? ib >= kb : ib <= kb) a[buf][ib] = UT ? ub[(4 * step) * LD + ib * 16] : ub[(ib * 16) * LD + 4 * step];
  };
#pragma unroll
  for (int ib = 0; ib < NT; ++ib) acc[ib] = d4{0, 0, 0, 0};
  fetch(0, 0);
  fetch(1, 1);
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    if (step + 2 < NSTEP) fetch((step + 2) % 3, step + 2);
    __builtin_amdgcn_sched_barrier(0);                 // (keeps the requests ahead of the products in the instruction stream)
    const int kb = step >> 2;
#pragma unroll
    for (int ib = 0; ib < NT; ++ib)
      if (UT ? ib >= kb : ib <= kb) acc[ib] = mfma(a[step % 3][ib], b[step % 3], acc[ib]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int SET>
struct SyrkSet {   // the 15 lower tiles over two waves.  SET 0: rows 4 and 3 (9 tiles) | SET 1: rows 2, 1, 0 (6 tiles)
  static constexpr int nt = SET == 0 ? 9 : 6;
  static constexpr int nv = SET == 0 ? 5 : 3;
  static constexpr int ib(int q) { return SET == 0 ? (q < 5 ? 4 : 3) : (q < 3 ? 2 : (q < 5 ? 1 : 0)); }
  static constexpr int jb(int q) { return SET == 0 ? (q < 5 ? q : q - 5) : (q < 3 ? q : (q < 5 ? q - 3 : 0)); }
};
template <int SET>
__device__ __forceinline__ void syrk_load(d4 (&acc)[9], const double* __restrict__ Ag, bool load, int li, int lk) {
  using T = SyrkSet<SET>;
  const double* base = Ag + lk * BS + li;
#pragma unroll
  for (int q = 0; q < T::nt; ++q) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[q][rr] = load ? base[(T::ib(q) * 16 + 4 * rr) * BS + T::jb(q) * 16] : 0.0;
  }
}
template <int SET>
__device__ __forceinline__ void syrk_run(d4 (&acc)[9], const double* W, double* __restrict__ Ag, int li, int lk) {
  using T = SyrkSet<SET>;
  const double* p = W + lk * LD + li;
  double v0[T::nv], v1[T::nv];
#pragma unroll
  for (int cidx = 0; cidx < T::nv; ++cidx) v0[cidx] = p[cidx * 16];
#pragma unroll
  for (int s = 0; s < BS / 4; s += 2) {      // operands of step s + 1 requested before the matrix-core work of step s
#pragma unroll
    for (int cidx = 0; cidx < T::nv; ++cidx) v1[cidx] = p[(4 * (s + 1)) * LD + cidx * 16];
#pragma unroll
    for (int q = 0; q < T::nt; ++q) acc[q] = mfma(-v0[T::ib(q)], v0[T::jb(q)], acc[q]);
    if (s + 2 < BS / 4) {
#pragma unroll
      for (int cidx = 0; cidx < T::nv; ++cidx) v0[cidx] = p[(4 * (s + 2)) * LD + cidx * 16];
    }
#pragma unroll
    for (int q = 0; q < T::nt; ++q) acc[q] = mfma(-v1[T::ib(q)], v1[T::jb(q)], acc[q]);
  }
  double* base = Ag + lk * BS + li;
#pragma unroll
  for (int q = 0; q < T::nt; ++q) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) base[(T::ib(q) * 16 + 4 * rr) * BS + T::jb(q) * 16] = acc[q][rr];
  }
}

// G = U U^T: the 15 lower tiles dealt to the waves at compile time (tile t = WAVE + 4 q), so a wave's products are
// straight-line code whose operand reads and matrix-core chains interleave.
template <int WAVE>
__device__ __forceinline__ void gram_tiles(const double* U, double* G, int li, int lk) {
  constexpr int8_t TI[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4};
  constexpr int8_t TJ[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
  constexpr int nq = WAVE < 3 ? 4 : 3;
  d4 g[nq];
#pragma unroll
  for (int q = 0; q < nq; ++q) {
    constexpr int dummy = 0;
    (void)dummy;
    const int ib = TI[WAVE + 4 * q], jb = TJ[WAVE + 4 * q];
    g[q] = tile_u_ut(U, ib, jb, li, lk);
  }
#pragma unroll
  for (int q = 0; q < nq; ++q) {
    const int ib = TI[WAVE + 4 * q], jb = TJ[WAVE + 4 * q];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      G[(ib * 16 + lk + 4 * rr) * LD + jb * 16 + li] = g[q][rr];
      if (ib != jb) G[(jb * 16 + li) * LD + ib * 16 + lk + 4 * rr] = g[q][rr];
    }
  }
}

__global__ void __launch_bounds__(256)
k_chunk_sweep(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, int* numeric_err, const int* __restrict__ status,
               int m, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Xc = reinterpret_cast<double*>(smem_raw);   // D~_k -> U_k
  double* Xn = Xc + MAT;                               // G_k -> stencil workspace -> D~_k+1 -> U_k+1
  double* Y = Xn + MAT;                                // spike: F_k -> W -> T_k -> F_k+1; column 79 = right-hand side
  double* cL = Y + MAT;                                // coupling tables of the current node
  double* cR = cL + 9 * NP;
  double* bv = cR + 9 * NP;                            // [80] right-hand side of the node built last
  double* red = bv + BS;                               // [8]
  int* sync = reinterpret_cast<int*>(red + 8);
  double* kq = red + 16;                               // [25] each: copies of K.q_w, K.lo, K.hi
  double* klo = kq + NP;
  double* khi = klo + NP;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const FteConst& K = *cst;
  const int c = blockIdx.x;
  const int first = c * m;
  const bool hasL = c > 0, hasR = c + 1 < n_chunks;
  const int n_int = hasR ? m - 1 : ch.n_nodes - first;
  const size_t MB = (size_t)BS * BS;
  int t01 = 0, t23 = 0, tflag = 0;           // rounds of the three wave-subset barriers (sync[0], [1], [2])
  // (debug stamps: workgroup dbg[29] writes wall-clock ticks of its phases at node dbg[30]; selectors read ONCE)
  long long* const dbgp = (ch.dbg && (long long)blockIdx.x == ch.dbg[29]) ? ch.dbg : nullptr;
  const int dbg_k = dbgp ? (int)ch.dbg[30] : -1;
#define SW_STAMP(i) do { if (dbgp && k == dbg_k && lane == 0) dbgp[i] = (long long)wall_clock64(); } while (0)
  if (tid < 4) sync[tid] = 0;
  if (tid < NP) {
    kq[tid] = K.q_w[tid];
    klo[tid] = K.lo[tid];
    khi[tid] = K.hi[tid];
  }
  __syncthreads();

  {  // ---- first node of the run, its spike F_0 = E_l (dense form) with the right-hand side in column 79
    NodeFetch f;
    build_fetch(f, ch, K, first, tid);
    fill_coupling_coef(cL, cR, K, first, tid, kq);
    for (int e = tid; e < MAT; e += 256) Y[e] = 0.0;
    const double gmax = build_finish(Xc, bv, f, K, first, tid, kq, klo, khi);
    publish_gmax(gmax, red, ch.gn_part, first, tid);   // (barrier inside: node, bv, tables, zeros complete)
    if (hasL)
      for (int e = tid; e < 9 * NP; e += 256) {
        const int pair = e / NP, p = e % NP, ii = pair / 3, jj = pair % 3;
        if (ii <= jj) Y[(ii * NP + p) * LD + jj * NP + p] = cL[e];
      }
    if (tid < BS) Y[tid * LD + (BS - 1)] = bv[tid];
    __syncthreads();
    if (wave < 2) chol80_pair(Xc, wave, lane, numeric_err, sync, t01, tflag);
    __syncthreads();
  }

#pragma unroll 1
  for (int k = 0; k < n_int; ++k) {
    const int node = first + k, next = node + 1;
    const bool last = k + 1 == n_int;
    const bool has_next = !last || hasR;
    // ================= serial part: G_k, then the next node =================
    // (thread index made opaque per iteration: the index arithmetic of the build / stencil phases is recomputed instead of
    //  being hoisted out of the node loop and spilled - a scratch reload behind the 51 KB store of G waits for that store)
    const int tid_ = opaque(tid);
    const int sp_ = tid_ % NP, sc0 = tid_ / NP;
    const bool s_act = tid_ < 10 * NP;
    NodeFetch f;
    if (has_next) build_fetch(f, ch, K, next, tid_);
    if (wave == 0) SW_STAMP(0);
    if (k > 0) fill_coupling_coef(cL, cR, K, node, tid_, kq);
    {
      const int gi = opaque(li), gk = opaque(lk);
      if (wave == 0) gram_tiles<0>(Xc, Xn, gi, gk);
      else if (wave == 1) gram_tiles<1>(Xc, Xn, gi, gk);
      else if (wave == 2) gram_tiles<2>(Xc, Xn, gi, gk);
      else gram_tiles<3>(Xc, Xn, gi, gk);
    }
    __syncthreads();                                   // G in Xn, tables of this node visible
    if (wave == 0) SW_STAMP(1);
    if (has_next) {
      // the next node's H / g / x were requested at the top of the iteration and have arrived; pin them down HERE: behind
      // the 51 KB store of G the wait for them would be a wait for the stores as well (one counter for loads and stores)
      // ("+v": the values leave the asm as NEW definitions, so no later use is tied to the loads' counter any more)
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(f.hv[q]));
      asm volatile("" : "+v"(f.xv), "+v"(f.gv), "+v"(f.lam));
    }
    store_mat(ch.D + node * MB, Xn, tid_);
    if (has_next) {
      double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
      if (s_act) {
        c00 = cR[(0 * 3 + 0) * NP + sp_];  c01 = cR[(0 * 3 + 1) * NP + sp_];  c02 = cR[(0 * 3 + 2) * NP + sp_];
        c11 = cR[(1 * 3 + 1) * NP + sp_];  c12 = cR[(1 * 3 + 2) * NP + sp_];  c22 = cR[(2 * 3 + 2) * NP + sp_];
      }
      __syncthreads();                                 // the store above has read Xn
      if (s_act) {                                     // pass 1: Xn <- G E, in place (rows sc0 + 10 j, state sp_)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double* rowp = Xn + (sc0 + 10 * j) * LD + sp_;
          const double g0 = rowp[0], g1 = rowp[NP], g2 = rowp[2 * NP];
          rowp[0] = g0 * c00 + g1 * c01 + g2 * c02;
          rowp[NP] = g1 * c11 + g2 * c12;
          rowp[2 * NP] = g2 * c22;
        }
      }
      __syncthreads();
      double dv[8][3];
      if (wave == 0) SW_STAMP(2);
      if (s_act) {                                     // pass 2: dv = -E^T (G E) (columns sc0 + 10 j, state sp_)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double* colp = Xn + sp_ * LD + sc0 + 10 * j;
          const double t0 = colp[0], t1 = colp[NP * LD], t2 = colp[2 * NP * LD];
          dv[j][0] = -(c00 * t0 + c01 * t1 + c02 * t2);
          dv[j][1] = -(c11 * t1 + c12 * t2);
          dv[j][2] = -(c22 * t2);
        }
      }
      __syncthreads();                                 // pass-2 reads done: Xn may be rebuilt
      if (wave == 0) SW_STAMP(3);
      const double gmax = build_finish(Xn, bv, f, K, next, tid_, kq, klo, khi,
                                       (dbgp && k == dbg_k) ? dbgp : nullptr);
      if (wave == 0) SW_STAMP(13);
      publish_gmax(gmax, red, ch.gn_part, next, tid_);  // (barrier inside)
      if (wave == 0) SW_STAMP(14);
      if (s_act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double* colp = Xn + sp_ * LD + sc0 + 10 * j;
          colp[0] += dv[j][0];
          colp[NP * LD] += dv[j][1];
          colp[2 * NP * LD] += dv[j][2];
        }
      }
    }
    if (wave == 0) SW_STAMP(15);
    __syncthreads();
    if (wave == 0) SW_STAMP(4);
    // ================= parallel part =================
    // waves 0, 1: blocked Cholesky of the next node (pivot chains | panel and trailing tiles)
    // waves 2, 3: W = U_k^T F_k on their 16-column strips, then D_L -= W^T W
    if (wave < 2) {
      if (!last)
        chol80_pair(Xn, wave, opaque(lane), numeric_err, sync, t01, tflag,
                    (dbgp && k == dbg_k) ? dbgp : nullptr);
      if (wave == 0) SW_STAMP(5);
    } else {
      const int li = opaque(lane & 15), lk = opaque(lane >> 4);
      if (hasL) {
        // the left separator's update so far (HBM / L2, owned by this workgroup): requested now, needed after W
        d4 accL[9];
        double* Ag = opaque_ptr(sp.AL + (size_t)(c - 1) * MB);
        if (wave == 2) {
          syrk_load<1>(accL, Ag, k > 0, li, lk);
          const int jbs[3] = {0, 1, 4};
          strips_ut_f<3>(Xc, Y, jbs, li, lk);
        } else {
          syrk_load<0>(accL, Ag, k > 0, li, lk);
          const int jbs[2] = {2, 3};
          strips_ut_f<2>(Xc, Y, jbs, li, lk);
        }
        SW_STAMP(8 + 8 * (wave - 2));
        sub_barrier(sync + 1, t23, 2, lane);           // every strip of W is in Y
        SW_STAMP(9 + 8 * (wave - 2));
        if (wave == 2) syrk_run<1>(accL, Y, Ag, li, lk);
        else syrk_run<0>(accL, Y, Ag, li, lk);
        SW_STAMP(10 + 8 * (wave - 2));
      } else if (wave == 2) {                          // first run: only the right-hand side column is alive
        const int jbs[1] = {4};
        strips_ut_f<1>(Xc, Y, jbs, li, lk);
      }
    }
    __syncthreads();                                   // next node factored; W complete and no longer needed by W^T W
    if (wave == 0) SW_STAMP(6);
    {
      // ---- T = U_k W, all four waves: wave w takes strip w, strip 4 (11 columns + the right-hand side) is dealt by row
      //      tile: (0,4) | (1,4) | (2,4), (4,4) | (3,4).  Results wait in registers until every read of W is done.
      const int li = opaque(lane & 15), lk = opaque(lane >> 4);
      d4 town[1][NT], tx0 = {0, 0, 0, 0}, tx1 = {0, 0, 0, 0};
      const int jbs[1] = {hasL ? wave : 4};
      const bool own = hasL || wave == 2;
      if (own) strips_u_w_acc<1>(Xc, Y, jbs, town, li, lk);
      if (hasL) {
        if (wave == 0) tx0 = tile_u_w(Xc, Y, 0, 4, li, lk);
        else if (wave == 1) tx0 = tile_u_w(Xc, Y, 1, 4, li, lk);
        else if (wave == 2) {
          tx0 = tile_u_w(Xc, Y, 2, 4, li, lk);
          tx1 = tile_u_w(Xc, Y, 4, 4, li, lk);
        } else tx0 = tile_u_w(Xc, Y, 3, 4, li, lk);
      }
      __syncthreads();
      if (own) {
#pragma unroll
        for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, jbs[0], town[0][ib], li, lk);
      }
      if (hasL) {
        tile_store(Y, wave == 0 ? 0 : (wave == 1 ? 1 : (wave == 2 ? 2 : 3)), 4, tx0, li, lk);
        if (wave == 2) tile_store(Y, 4, 4, tx1, li, lk);
      }
    }
    __syncthreads();                                   // T_k complete in Y
    if (wave == 0) SW_STAMP(7);
    {
      // ---- T^T -> HBM (the back-substitution reads along its columns; column 79 is z_k), then F_k+1 = -E^T T_k in place,
      //      column 79 += the next node's right-hand side.  20 columns per wave (first run: column 79 only).
      const int c_lo = hasL ? 20 * wave : BS - 1, c_hi = hasL ? c_lo + 20 : (wave == 3 ? BS : BS - 1);
      if (hasL) {
        double* Tg = ch.Wl + node * MB;
        // (all 25 LDS reads of the wave first, then the stores)
        double v0[20], v1[5];
#pragma unroll
        for (int j = 0; j < 20; ++j) v0[j] = Y[lane * LD + c_lo + j];
#pragma unroll
        for (int j = 0; j < 5; ++j) v1[j] = Y[(64 + (lane & 15)) * LD + c_lo + 5 * (lane >> 4) + j];   // rows 64..79: 4 x 5 columns
#pragma unroll
        for (int j = 0; j < 20; ++j) Tg[(size_t)(c_lo + j) * BS + lane] = v0[j];
#pragma unroll
        for (int j = 0; j < 5; ++j) Tg[(size_t)(c_lo + 5 * (lane >> 4) + j) * BS + 64 + (lane & 15)] = v1[j];
      }
      if (wave == 3) {
        ch.b[(size_t)node * BS + lane] = Y[lane * LD + (BS - 1)];
        if (lane < 16) ch.b[(size_t)node * BS + 64 + lane] = Y[(64 + lane) * LD + (BS - 1)];
      }
      if (has_next) {
        // lane = (state p, column group): the six stencil coefficients of p are read once, rows p, 25 + p, 50 + p of the
        // lane's columns are rewritten in place
        const int ncol = c_hi - c_lo, p = lane % NP, cg = lane / NP;        // cg 0, 1 (lanes 50..63 idle)
        if (cg < 2 && ncol > 0) {
          const double e00 = cR[(0 * 3 + 0) * NP + p], e01 = cR[(0 * 3 + 1) * NP + p], e02 = cR[(0 * 3 + 2) * NP + p];
          const double e11 = cR[(1 * 3 + 1) * NP + p], e12 = cR[(1 * 3 + 2) * NP + p], e22 = cR[(2 * 3 + 2) * NP + p];
          const double b0 = bv[p], b1 = bv[NP + p], b2 = bv[2 * NP + p];
          for (int cc = c_lo + cg; cc < c_hi; cc += 2) {
            const double f0 = Y[p * LD + cc], f1 = Y[(NP + p) * LD + cc], f2 = Y[(2 * NP + p) * LD + cc];
            double o0 = -(e00 * f0 + e01 * f1 + e02 * f2), o1 = -(e11 * f1 + e12 * f2), o2 = -(e22 * f2);
            if (cc == BS - 1) {
              o0 += b0;
              o1 += b1;
              o2 += b2;
            }
            Y[p * LD + cc] = o0;
            Y[(NP + p) * LD + cc] = o1;
            Y[(2 * NP + p) * LD + cc] = o2;
          }
        }
        if (lane < ncol)
          for (int r = 3 * NP; r < BS; ++r) Y[r * LD + c_lo + lane] = 0.0;     // padding rows couple to nothing
      }
    }
    __syncthreads();
    if (wave == 0) SW_STAMP(27);
    if (last && hasR) {                                // the node built last is the right separator
      store_mat(sp.D + (size_t)c * MB, Xn, tid);
      if (tid < BS) sp.b[(size_t)c * BS + tid] = Y[tid * LD + (BS - 1)];
      if (hasL) {
        double* Cg = sp.Cpl + (size_t)(c - 1) * MB;    // block(R, L): rows R, columns L
        for (int e = tid; e < BS * BS; e += 256) {
          const int r = e / BS, cc = e % BS;
          Cg[e] = cc < 3 * NP ? Y[r * LD + cc] : 0.0;
        }
      }
    }
    double* tmp = Xc;
    Xc = Xn;
    Xn = tmp;
  }
#undef SW_STAMP
}

// ================================================================================================================
// The sweep kernel, eight waves.  The node loop has two dependency chains that only meet once per node:
//   D chain     : U_k -> G_k = U_k U_k^T -> D~_k+1 = D_k+1 - E^T G_k E -> Cholesky -> U_k+1            (Xc -> Xn)
//   spike chain : F_k -> W = U_k^T F_k -> D_L -= W^T W,  T_k = U_k W -> HBM,  F_k+1 = -E^T T_k          (Xc, Y)
// Per node: a SERIAL part on all eight waves (G_k, its store, the two stencil passes, the next node built), then a
// PARALLEL part in which waves 0..2 factor the next node (wave 0: the five 16-pivot chains with the look-ahead tile;
// waves 1, 2: panel and trailing tiles, alternate products) while waves 3..7 run the WHOLE spike chain of node k, each
// on its own 16-column strip of the spike (strip = wave - 3; only W^T W reads across strips: two 5-wave LDS-counter
// barriers per node).  Two waves share every SIMD, so the matrix-core work of the spike fills the issue slots the
// pivot chains leave empty.  Same arithmetic, tile by tile, as the four-wave form it replaces.
constexpr int SW8_T = 512;
// role of wave w in the parallel part: 0 = pivot chains, 1 / 2 = factor helpers, 3 + sw = spike strip sw.  Waves w and w + 4
// share a SIMD and its matrix pipe; the matrix-core work is dealt so that the four pipes carry about the same load:
//   SIMD 0: pivot chains (~70 matrix instructions, at raised priority) + strip 0 (180)
//   SIMD 1, 2: one factor helper each (~100) + strips 1, 2 (180 each)     SIMD 3: strips 3 and 4 (360)
__device__ __forceinline__ int role8(int wave) { return (0x75436210u >> (4 * wave)) & 15; }   // {0, 1, 2, 6, 3, 4, 5, 7}
// G = U U^T: the 15 lower tiles dealt by matrix-core work (tile (ib, jb) costs 4 (5 - ib) instructions): wave w takes the
// tiles in the nibbles of entry w, 15 = none:  {0} {1,10} {2,11} {3,6} {4,7} {5,8} {9,12,13} {14}
__device__ __forceinline__ int gram8_tile(int wave, int q) {
  const unsigned long long tb0 = 0x0F630FB20FA10FF0ull;   // waves 0..3, 16 bits each, nibble q = q-th tile
  const unsigned long long tbl = 0x0FFE0DC90F850F74ull;   // waves 4..7
  const unsigned v = (unsigned)((wave < 4 ? tb0 : tbl) >> (16 * (wave & 3))) & 0xFFFFu;
  const int t = (v >> (4 * q)) & 15;
  return t == 15 ? -1 : t;
}

// chol80 by THREE waves: role 0 = pivot chains + look-ahead (never waits for a helper inside a step), roles 1, 2 = the
// other panel tiles (2 + 1) and the trailing products (alternate entries of the step's list).
// sync[0]: barrier of the three, sync[2]: barrier of the two helpers, sync[3]: one-way flag "role 0's panel tile posted".
__device__ __forceinline__ void chol80_trio(double* Lm, int role, int lane, int* err, int* sync, int& t3, int& t2, int& posted,
                                            long long* dbg = nullptr) {
  const int li = lane & 15, lk = lane >> 4;
  if (role == 0) {
    __builtin_amdgcn_s_setprio(3);                     // the chain's VALU wins the issue arbitration against its SIMD mate
    chol16_inv(Lm, lane, err);
  }
  sub_barrier(sync, t3, 3, lane);
#pragma unroll 1
  for (int kb = 0; kb < NT; ++kb) {
    {  // panel: tile(t, kb) <- tile(t, kb) U_kk
      const double* Ukk = Lm + (kb * 16) * LD + kb * 16;
      double bq[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bq[s] = Ukk[(4 * s + lk) * LD + li];
      if (role != 1) {
        double* A = Lm + (panel_tile(kb, role == 0 ? 0 : 3) * 16) * LD + kb * 16;
        double av[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = A[li * LD + 4 * s + lk];
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(av[s], bq[s], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) A[(lk + 4 * rr) * LD + li] = acc[rr];
      } else {
        double av[2][4];
        d4 acc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double* A = Lm + (panel_tile(kb, q + 1) * 16) * LD + kb * 16;
#pragma unroll
          for (int s = 0; s < 4; ++s) av[q][s] = A[li * LD + 4 * s + lk];
          acc[q] = d4{0, 0, 0, 0};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[q] = mfma(av[q][s], bq[s], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          double* A = Lm + (panel_tile(kb, q + 1) * 16) * LD + kb * 16;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) A[(lk + 4 * rr) * LD + li] = acc[q][rr];
        }
      }
    }
    if (kb == NT - 1) {
      if (role == 0) __builtin_amdgcn_s_setprio(0);
      sub_barrier(sync, t3, 3, lane);
      break;
    }
    ++posted;
    if (role == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (dbg && lane == 0) dbg[40 + 2 * kb] = (long long)wall_clock64();
      // next diagonal tile, then its 16-pivot chain
      double* Cc = Lm + ((kb + 1) * 16) * LD + (kb + 1) * 16;
      const double* A = Lm + ((kb + 1) * 16) * LD + kb * 16;
      d4 a;
      double av[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(lk + 4 * rr) * LD + li];
#pragma unroll
      for (int s = 0; s < 4; ++s) av[s] = A[li * LD + 4 * s + lk];
#pragma unroll
      for (int s = 0; s < 4; ++s) a = mfma(-av[s], av[s], a);
      chol16_inv_acc(Cc, a, lane, err);
      if (dbg && lane == 0) dbg[41 + 2 * kb] = (long long)wall_clock64();
    } else {
      sub_barrier(sync + 2, t2, 2, lane);              // the helpers' three panel tiles
      while (__hip_atomic_load(sync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < posted) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (role == 1) {
        if (kb == 0) trail_step<0, 0, 2>(Lm, li, lk);
        else if (kb == 1) trail_step<1, 0, 2>(Lm, li, lk);
        else if (kb == 2) trail_step<2, 0, 2>(Lm, li, lk);
        else trail_step<3, 0, 2>(Lm, li, lk);
      } else {
        if (kb == 0) trail_step<0, 1, 2>(Lm, li, lk);
        else if (kb == 1) trail_step<1, 1, 2>(Lm, li, lk);
        else if (kb == 2) trail_step<2, 1, 2>(Lm, li, lk);
        else trail_step<3, 1, 2>(Lm, li, lk);
      }
      if (dbg && lane == 0) dbg[48 + 4 * (role - 1) + kb] = (long long)wall_clock64();
    }
    sub_barrier(sync, t3, 3, lane);
  }
}

// The NQ tiles (ib[q], jb[q]) of the left separator's update D_L -= W^T W that one spike wave owns: accumulators
// loaded from / stored to the workgroup's block in global memory (L2-resident between nodes).
template <int NQ>
__device__ __forceinline__ void syrk_load_n(d4 (&acc)[NQ], const double* __restrict__ Ag, bool load, const int (&ib)[NQ],
                                            const int (&jb)[NQ], int li, int lk) {
  // (loads unconditional, THEN the select: a conditional load compiles to a branch per element with a full wait in front)
  const double* base = Ag + lk * BS + li;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[q][rr] = base[(ib[q] * 16 + 4 * rr) * BS + jb[q] * 16];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[q][rr] = load ? acc[q][rr] : 0.0;
}
template <int NQ>
__device__ __forceinline__ void syrk_run_n(d4 (&acc)[NQ], const double* W, double* __restrict__ Ag, const int (&ib)[NQ],
                                           const int (&jb)[NQ], int li, int lk) {
  const double* p = W + lk * LD + li;
  double a0[NQ], b0[NQ], a1[NQ], b1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    a0[q] = p[ib[q] * 16];
    b0[q] = p[jb[q] * 16];
  }
#pragma unroll
  for (int s = 0; s < BS / 4; s += 2) {      // operands of step s + 1 requested before the matrix-core work of step s
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      a1[q] = p[(4 * (s + 1)) * LD + ib[q] * 16];
      b1[q] = p[(4 * (s + 1)) * LD + jb[q] * 16];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = mfma(-a0[q], b0[q], acc[q]);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < BS / 4) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        a0[q] = p[(4 * (s + 2)) * LD + ib[q] * 16];
        b0[q] = p[(4 * (s + 2)) * LD + jb[q] * 16];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = mfma(-a1[q], b1[q], acc[q]);
    __builtin_amdgcn_sched_barrier(0);
  }
  double* base = Ag + lk * BS + li;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) base[(ib[q] * 16 + 4 * rr) * BS + jb[q] * 16] = acc[q][rr];
}

__global__ void __launch_bounds__(SW8_T)
k_chunk_sweep8(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, int* numeric_err, const int* __restrict__ status,
               int m, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Xc = reinterpret_cast<double*>(smem_raw);   // D~_k -> U_k
  double* Xn = Xc + MAT;                               // G_k -> stencil workspace -> D~_k+1 -> U_k+1
  double* Y = Xn + MAT;                                // spike: F_k -> W -> T_k -> F_k+1; column 79 = right-hand side
  double* cL = Y + MAT;                                // coupling tables of the current node
  double* cR = cL + 9 * NP;
  double* bv = cR + 9 * NP;                            // [80] right-hand side of the node built last
  double* red = bv + BS;                               // [8]
  int* sync = reinterpret_cast<int*>(red + 8);         // [0] factor trio | [1] spike waves | [2] helper pair | [3] flag
  double* kq = red + 16;                               // [25] each: copies of K.q_w, K.lo, K.hi
  double* klo = kq + NP;
  double* khi = klo + NP;
  long long* lst = reinterpret_cast<long long*>(khi + NP);   // [64] debug stamps, copied out at the end of the stamped node
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const FteConst& K = *cst;
  const int c = blockIdx.x;
  const int first = c * m;
  const bool hasL = c > 0, hasR = c + 1 < n_chunks;
  const int n_int = hasR ? m - 1 : ch.n_nodes - first;
  const size_t MB = (size_t)BS * BS;
  const int role = __builtin_amdgcn_readfirstlane(role8(wave));
  const int n_spike = hasL ? 5 : 1;          // first run: only the right-hand side column (strip 4) is alive
  int t3 = 0, t2 = 0, tflag = 0, t5 = 0;     // rounds of the wave-subset barriers
  // (debug stamps: workgroup dbg[29] writes wall-clock ticks of its phases at node dbg[30]; selectors read ONCE)
  long long* const dbgp = (ch.dbg && (long long)blockIdx.x == ch.dbg[29]) ? ch.dbg : nullptr;
  const int dbg_k = dbgp ? (int)ch.dbg[30] : -1;
  // (stamps go to LDS: a global store in front of a release fence would itself delay the wave that is being timed)
#define SW_STAMP(i) do { if (dbgp && k == dbg_k && lane == 0) lst[i] = (long long)wall_clock64(); } while (0)
  if (tid < 4) sync[tid] = 0;
  if (tid < 64) lst[tid] = 0;
  if (tid < NP) {
    kq[tid] = K.q_w[tid];
    klo[tid] = K.lo[tid];
    khi[tid] = K.hi[tid];
  }
  __syncthreads();

  {  // ---- first node of the run, its spike F_0 = E_l (dense form) with the right-hand side in column 79
    NodeFetch f;
    build_fetch<SW8_T>(f, ch, K, first, tid);
    fill_coupling_coef<SW8_T>(cL, cR, K, first, tid, kq);
    for (int e = tid; e < MAT; e += SW8_T) Y[e] = 0.0;
    const double gmax = build_finish<SW8_T>(Xc, bv, f, K, first, tid, kq, klo, khi);
    publish_gmax<SW8_T>(gmax, red, ch.gn_part, first, tid);   // (barrier inside: node, bv, tables, zeros complete)
    if (hasL)
      for (int e = tid; e < 9 * NP; e += SW8_T) {
        const int pair = e / NP, p = e % NP, ii = pair / 3, jj = pair % 3;
        if (ii <= jj) Y[(ii * NP + p) * LD + jj * NP + p] = cL[e];
      }
    if (tid < BS) Y[tid * LD + (BS - 1)] = bv[tid];
    __syncthreads();
    if (role < 3) chol80_trio(Xc, role, lane, numeric_err, sync, t3, t2, tflag);
    __syncthreads();
  }

#pragma unroll 1
  for (int k = 0; k < n_int; ++k) {
    const int node = first + k, next = node + 1;
    const bool last = k + 1 == n_int;
    const bool has_next = !last || hasR;
    // ================= serial part (all eight waves): G_k, then the next node =================
    // (thread index made opaque per iteration: the index arithmetic of the build / stencil phases is recomputed instead of
    //  being hoisted out of the node loop and spilled - a scratch reload behind the 51 KB store of G waits for that store)
    const int tid_ = opaque(tid);
    // The next node is built by PAIRS OF STATES: thread e (and e + 512 < 625) owns the 3 x 3 frame block of the state pair
    // (p, p') = (e / 25, e % 25) - the nine entries [(j, p)][(j', p')] of the node.  Everything that lands in the block is the
    // thread's own: the Gauss-Newton entries H_j[p][p'] (requested here, a whole node ahead of their use), the Schur update
    // -(E^T G_k E) of the block (nine entries of G_k in, nine out: E couples only equal states), and for p = p' the damping,
    // the bound pinning, the intra-node third-difference couplings and the right-hand side.  No zero fill, no in-place pass.
    const int fb_next = 3 * next;                      // (chunked contexts have no pinned separator: node t holds frames 3t ..)
    double hq[2][3] = {{0, 0, 0}, {0, 0, 0}}, xq[3] = {0, 0, 0}, gq[3] = {0, 0, 0}, cvq[3] = {0, 0, 0}, lamq = 0.0;
    bool liveq[3] = {false, false, false}, ownq[3] = {false, false, false};
    const int e1 = tid_ + SW8_T;
    const bool own1 = e1 < NP * NP;
    const int pd0 = tid_ / NP, pc0 = tid_ % NP, pd1 = own1 ? e1 / NP : 0, pc1 = own1 ? e1 % NP : 0;
    const bool diag0 = pd0 == pc0, diag1 = own1 && pd1 == pc1;
    const int pdg = diag1 ? pd1 : pd0;                 // (a thread owns at most one diagonal pair)
    if (has_next) {
      const int cur = ch.st->cur;
      const double* Hg = cur ? ch.H1 : ch.H0;
      lamq = ch.st->lam;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (fb_next + j < K.n_frames) {
          hq[0][j] = Hg[(size_t)(fb_next + j) * NP * NP + tid_];
          if (own1) hq[1][j] = Hg[(size_t)(fb_next + j) * NP * NP + e1];
        }
      if (diag0 || diag1) {
        const double* xg = cur ? ch.x1 : ch.x0;
        const double* gg = cur ? ch.g1 : ch.g0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (fb_next + j < K.n_frames) {
            xq[j] = xg[(size_t)(fb_next + j + HALO) * NP + pdg];
            gq[j] = gg[(size_t)(fb_next + j) * NP + pdg];
          }
        // intra-node third-difference couplings of state pdg, frame pairs (0,1), (0,2), (1,2) (needed when the node is written:
        // worked out here, beside the loads in flight, so that the write phase reads no constants through the scalar cache)
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          const int j = pr == 2 ? 1 : 0, jp = pr == 0 ? 1 : 2;
          if (fb_next + jp < K.n_frames)
            cvq[pr] = 2.0 * kq[pdg] * band_coef_clip(K.n_offset + fb_next + j, jp - j, K.n_global, K.clip_len);
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        liveq[j] = fb_next + j < K.n_frames;
        ownq[j] = fb_next + j >= K.own_lo && fb_next + j < K.own_hi;
      }
    }
    if (wave == 0) SW_STAMP(0);
    {
      const int gi = opaque(lane & 15), gk = opaque(lane >> 4);
      d4 g[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int t = gram8_tile(wave, q);
        if (t >= 0) g[q] = tile_u_ut(Xc, tri_i(t), tri_j(t), gi, gk);
      }
      if (k > 0) fill_coupling_coef<SW8_T>(cL, cR, K, node, tid_, kq);   // (beside the matrix-core work above)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int t = gram8_tile(wave, q);
        if (t >= 0) {
          const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            Xn[(ib * 16 + gk + 4 * rr) * LD + jb * 16 + gi] = g[q][rr];
            if (ib != jb) Xn[(jb * 16 + gi) * LD + ib * 16 + gk + 4 * rr] = g[q][rr];
          }
        }
      }
    }
    __syncthreads();                                   // G in Xn, tables of this node visible
    if (wave == 0) SW_STAMP(1);
    // the next node's H / g / x were requested at the top of the iteration and have arrived; pin them down HERE: behind
    // the 51 KB store of G the wait for them would be a wait for the stores as well (one counter for loads and stores).
    // (Unconditionally: with a path around the pin the compiler's wait-count analysis still sees the loads pending below.)
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(hq[0][j]), "+v"(hq[1][j]), "+v"(xq[j]), "+v"(gq[j]));
    asm volatile("" : "+v"(lamq));
    {
      double2* d2 = reinterpret_cast<double2*>(ch.D + node * MB);
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int idx = tid_ + SW8_T * q;
        if (idx < BS * BS / 2) {
          const int e = 2 * idx, r = e / BS, cc = e % BS;
          d2[idx] = make_double2(Xn[r * LD + cc], Xn[r * LD + cc + 1]);
        }
      }
    }
    if (has_next) {
      // S = E^T G_k E on the thread's blocks: (G E)[(jj, p)][(i', p')] = sum_{jj' >= i'} G[(jj, p)][(jj', p')] c_{i' jj'}(p'),
      // S[(i, p)][(i', p')] = sum_{jj >= i} c_{i jj}(p) (G E)[(jj, p)][(i', p')]      (c = cR: the right coupling of node k)
      double S[2][3][3];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int pr = sl ? pd1 : pd0, pc = sl ? pc1 : pc0;
        if (sl == 0 || own1) {
          const double a00 = cR[0 * NP + pr], a01 = cR[1 * NP + pr], a02 = cR[2 * NP + pr];
          const double a11 = cR[4 * NP + pr], a12 = cR[5 * NP + pr], a22 = cR[8 * NP + pr];
          const double b00 = cR[0 * NP + pc], b01 = cR[1 * NP + pc], b02 = cR[2 * NP + pc];
          const double b11 = cR[4 * NP + pc], b12 = cR[5 * NP + pc], b22 = cR[8 * NP + pc];
          double m[3][3];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            const double* gp = Xn + (jj * NP + pr) * LD + pc;
            const double g0 = gp[0], g1 = gp[NP], g2 = gp[2 * NP];
            m[jj][0] = g0 * b00 + g1 * b01 + g2 * b02;
            m[jj][1] = g1 * b11 + g2 * b12;
            m[jj][2] = g2 * b22;
          }
#pragma unroll
          for (int ii = 0; ii < 3; ++ii) {
            S[sl][0][ii] = a00 * m[0][ii] + a01 * m[1][ii] + a02 * m[2][ii];
            S[sl][1][ii] = a11 * m[1][ii] + a12 * m[2][ii];
            S[sl][2][ii] = a22 * m[2][ii];
          }
        }
      }
      __syncthreads();                                 // every read of G done (the store above included): Xn is rebuilt
      if (wave == 0) SW_STAMP(2);
      double gmax = 0.0;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int pr = sl ? pd1 : pd0, pc = sl ? pc1 : pc0;
        const bool dg = sl ? diag1 : diag0;
        if (sl == 0 || own1) {
          double v[3][3];
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) v[j][jp] = (j == jp ? hq[sl][j] : 0.0);
          if (dg) {
            // diagonal pair: Marquardt damping, bound pinning, right-hand side, projected-gradient norm of rows (j, p) and the
            // intra-node third-difference couplings (frame pairs (0,1), (0,2), (1,2) of state p)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              double d = 1.0, bb = 0.0;
              if (liveq[j]) {
                d = hq[sl][j];
                const double gtol = GRAD_ZERO_REL * d;
                const bool fixed = (xq[j] <= klo[pr] && gq[j] > gtol) || (xq[j] >= khi[pr] && gq[j] < -gtol);
                d = d + lamq * fmax(d, DIAG_FLOOR);
                if (fixed) d *= FIX_SCALE;
                bb = fixed ? 0.0 : -gq[j];
                if (ownq[j]) gmax = fmax(gmax, fabs(bb));   // (window sharding: owned frames only)
              }
              v[j][j] = d;
              bv[j * NP + pr] = bb;
            }
            v[0][1] = v[1][0] = cvq[0];
            v[0][2] = v[2][0] = cvq[1];
            v[1][2] = v[2][1] = cvq[2];
          }
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) Xn[(j * NP + pr) * LD + jp * NP + pc] = v[j][jp] - S[sl][j][jp];
        }
      }
      // padding rows / columns 75 .. 79: identity
      for (int e = tid_; e < BS * BS - 9 * NP * NP; e += SW8_T) {
        const int r = e < 5 * BS ? 3 * NP + e / BS : (e - 5 * BS) / 5, cc = e < 5 * BS ? e % BS : 3 * NP + (e - 5 * BS) % 5;
        Xn[r * LD + cc] = r == cc ? 1.0 : 0.0;
      }
      if (tid_ < BS - 3 * NP) bv[3 * NP + tid_] = 0.0;
      if (wave == 0) SW_STAMP(3);
      publish_gmax<SW8_T>(gmax, red, ch.gn_part, next, tid_);  // (barrier inside)
    }
    __syncthreads();
    if (wave == 0) SW_STAMP(4);
    // ================= parallel part =================
    if (role < 3) {
      // blocked Cholesky of the next node
      if (!last)
        chol80_trio(Xn, role, opaque(lane), numeric_err, sync, t3, t2, tflag, (dbgp && k == dbg_k) ? lst : nullptr);
      SW_STAMP(8 + wave);
    } else if (hasL || role == 7) {
      // the spike chain of node k on strip sw (columns 16 sw .. 16 sw + 15)
      // (wave and lane indices made opaque per node: see tid_ above)
      const int sw = __builtin_amdgcn_readfirstlane(opaque(role)) - 3;
      const int ln = opaque(lane);
      const int li = ln & 15, lk = ln >> 4;
      // D_L -= W^T W: three of the 15 lower tiles per wave (tile t = sw + 5 q)
      d4 accL[3] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
      int sib[3], sjb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        sib[q] = tri_i(sw + 5 * q);
        sjb[q] = tri_j(sw + 5 * q);
      }
      double* Ag = sp.AL + (size_t)(hasL ? opaque(c - 1) : 0) * MB;   // (global address space kept: no pointer through asm)
      // the left separator's update so far (HBM / L2, owned by this workgroup): requested now, needed after W
      if (hasL) syrk_load_n<3>(accL, Ag, k > 0, sib, sjb, li, lk);
      {
        d4 wacc[NT];
        strip_product<true>(Xc, Y, sw, wacc, li, lk);  // W strip, then in place over F (own strip: every read precedes)
#pragma unroll
        for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, sw, wacc[ib], li, lk);
      }
      SW_STAMP(16 + 4 * sw);
      sub_barrier(sync + 1, t5, n_spike, ln);          // every strip of W is in Y
      SW_STAMP(17 + 4 * sw);
      if (hasL) syrk_run_n<3>(accL, Y, Ag, sib, sjb, li, lk);
      SW_STAMP(18 + 4 * sw);
      d4 town[NT];
      strip_product<false>(Xc, Y, sw, town, li, lk);   // T strip = U W strip (reads its own strip of W only)
      sub_barrier(sync + 1, t5, n_spike, ln);          // nobody reads W any more
#pragma unroll
      for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, sw, town[ib], li, lk);
      SW_STAMP(19 + 4 * sw);
      // ---- T^T -> HBM (the back-substitution reads along its columns; column 79 is z_k), then F_k+1 = -E^T T_k in place,
      //      column 79 += the next node's right-hand side.  Own strip only: no barrier (LDS operations of a wave are ordered)
      const int c_lo = 16 * sw;
      if (hasL) {
        double* Tg = ch.Wl + node * MB;
        double v0[16], v1[4];
#pragma unroll
        for (int j = 0; j < 16; ++j) v0[j] = Y[ln * LD + c_lo + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) v1[j] = Y[(64 + li) * LD + c_lo + 4 * lk + j];   // rows 64..79: 4 x 4 columns
#pragma unroll
        for (int j = 0; j < 16; ++j) Tg[(size_t)(c_lo + j) * BS + ln] = v0[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) Tg[(size_t)(c_lo + 4 * lk + j) * BS + 64 + li] = v1[j];
      }
      if (sw == 4) {
        ch.b[(size_t)node * BS + ln] = Y[ln * LD + (BS - 1)];
        if (ln < 16) ch.b[(size_t)node * BS + 64 + ln] = Y[(64 + ln) * LD + (BS - 1)];
      }
      if (has_next) {
        // lane = (state p, column group): the six stencil coefficients of p are read once, rows p, 25 + p, 50 + p of the
        // lane's columns are rewritten in place
        const int p = ln % NP, cg = ln / NP;        // cg 0, 1 (lanes 50..63 idle)
        if (cg < 2) {
          const double e00 = cR[(0 * 3 + 0) * NP + p], e01 = cR[(0 * 3 + 1) * NP + p], e02 = cR[(0 * 3 + 2) * NP + p];
          const double e11 = cR[(1 * 3 + 1) * NP + p], e12 = cR[(1 * 3 + 2) * NP + p], e22 = cR[(2 * 3 + 2) * NP + p];
          const double b0 = bv[p], b1 = bv[NP + p], b2 = bv[2 * NP + p];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int cc = c_lo + cg + 2 * q;
            const double f0 = Y[p * LD + cc], f1 = Y[(NP + p) * LD + cc], f2 = Y[(2 * NP + p) * LD + cc];
            double o0 = -(e00 * f0 + e01 * f1 + e02 * f2), o1 = -(e11 * f1 + e12 * f2), o2 = -(e22 * f2);
            if (cc == BS - 1) {
              o0 += b0;
              o1 += b1;
              o2 += b2;
            }
            Y[p * LD + cc] = o0;
            Y[(NP + p) * LD + cc] = o1;
            Y[(2 * NP + p) * LD + cc] = o2;
          }
        }
        if (ln < 16)
          for (int r = 3 * NP; r < BS; ++r) Y[r * LD + c_lo + ln] = 0.0;     // padding rows couple to nothing
      }
      SW_STAMP(8 + wave);
    }
    __syncthreads();                                   // next node factored, F_k+1 complete
    if (wave == 0) SW_STAMP(7);
    if (dbgp && k == dbg_k) {
      __syncthreads();
      if (tid < 64 && lst[tid]) dbgp[tid] = lst[tid];
    }
    if (last && hasR) {                                // the node built last is the right separator
      {
        double2* d2 = reinterpret_cast<double2*>(sp.D + (size_t)c * MB);
        for (int idx = tid; idx < BS * BS / 2; idx += SW8_T) {
          const int e = 2 * idx, r = e / BS, cc = e % BS;
          d2[idx] = make_double2(Xn[r * LD + cc], Xn[r * LD + cc + 1]);
        }
      }
      if (tid < BS) sp.b[(size_t)c * BS + tid] = Y[tid * LD + (BS - 1)];
      if (hasL) {
        double* Cg = sp.Cpl + (size_t)(c - 1) * MB;    // block(R, L): rows R, columns L
        for (int e = tid; e < BS * BS; e += SW8_T) {
          const int r = e / BS, cc = e % BS;
          Cg[e] = cc < 3 * NP ? Y[r * LD + cc] : 0.0;
        }
      }
    }
    double* tmp = Xc;
    Xc = Xn;
    Xn = tmp;
  }
#undef SW_STAMP
}

// Separator q: D += AL (the run on its right; lower tiles - the factorisation reads no others).  The right-hand side rode as
// column 79 of the spike, so row 79 of AL holds -(sum W^T y) = the update of b, and rows / columns >= 75 of AL are not
// part of the Schur update.
__global__ void __launch_bounds__(256) k_sep_combine(SepView sp, const int* __restrict__ status) {
  if (status && *status != 0) return;
  const int q = blockIdx.x, tid = threadIdx.x;
  const size_t MB = (size_t)BS * BS;
  double* D = sp.D + q * MB;
  const double* A = sp.AL + q * MB;
  for (int e = tid; e < BS * BS; e += 256) {
    const int r = e / BS, cc = e % BS;
    if ((cc >> 4) <= (r >> 4) && r < 3 * NP && cc < 3 * NP) D[e] += A[e];
  }
  if (tid < 3 * NP) sp.b[(size_t)q * BS + tid] += A[(size_t)(BS - 1) * BS + tid];
}

// One workgroup per run, right to left: x_k = z_k - G_k (E_r(k) x_k+1) - T_k x_L.  G_k is symmetric and T_k is stored
// transposed, so thread (column r, third of the rows) reads consecutive addresses along a wave - no LDS staging; the next
// node's operands are requested before the current node's sums are reduced.
__global__ void __launch_bounds__(256)
k_chunk_backsub(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, const int* __restrict__ status, int m,
                int n_chunks) {
  if (status && *status != 0) return;
  __shared__ double xn[BS], xl[BS], vv[BS], ysc[3 * BS], cL[9 * NP], cR[9 * NP];
  const int tid = threadIdx.x;
  const int c = blockIdx.x, first = c * m;
  const bool hasL = c > 0, hasR = c + 1 < n_chunks;
  const int n_int = hasR ? m - 1 : ch.n_nodes - first;
  const size_t MB = (size_t)BS * BS;
  const int col = tid % BS, part = tid / BS, c0 = 27 * part, nc = part < 2 ? 27 : 26;
  if (tid < BS) {
    xl[tid] = hasL ? sp.b[(size_t)(c - 1) * BS + tid] : 0.0;
    const double xr = hasR ? sp.b[(size_t)c * BS + tid] : 0.0;
    xn[tid] = xr;
    if (hasR) ch.b[(size_t)(first + n_int) * BS + tid] = xr;      // the separator's solution joins the chain's vector
  }
  double g[27], t[27];
  auto fetch = [&](int node) {
    if (tid < 3 * BS) {
      const double* G = ch.D + node * MB;
      const double* Tt = ch.Wl + node * MB;
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        g[k] = k < nc ? G[(size_t)(c0 + k) * BS + col] : 0.0;
        t[k] = (hasL && k < nc) ? Tt[(size_t)(c0 + k) * BS + col] : 0.0;
      }
    }
  };
  fetch(first + n_int - 1);
  for (int k = n_int - 1; k >= 0; --k) {
    const int node = first + k;
    fill_coupling_coef(cL, cR, *cst, node, tid);
    const double zi = tid < BS ? ch.b[(size_t)node * BS + tid] : 0.0;
    __syncthreads();                                   // tables, xn of the previous round
    if (tid < BS) {
      double v = 0.0;
      if (tid < 3 * NP) {
        const int a = tid / NP, p = tid % NP;
        for (int ii = 0; ii <= a; ++ii) v += cR[(ii * 3 + a) * NP + p] * xn[ii * NP + p];   // (E_r x_k+1)[(a, p)]
      }
      vv[tid] = v;
    }
    __syncthreads();
    if (tid < 3 * BS) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int kk = 0; kk < 27; ++kk) {
        const int cc = c0 + (kk < nc ? kk : 0);
        s0 += g[kk] * vv[cc];
        s1 += t[kk] * xl[cc];
      }
      ysc[tid] = s0 + s1;
    }
    if (k > 0) fetch(node - 1);
    __syncthreads();
    if (tid < BS) {
      const double x = zi - ((ysc[tid] + ysc[BS + tid]) + ysc[2 * BS + tid]);
      xn[tid] = x;
      ch.b[(size_t)node * BS + tid] = x;
    }
  }
}

static bool sweep_eight_waves() {
  static const bool v = [] {
    const char* e = std::getenv("ACINO_SWEEP_WAVES");
    return !(e && std::atoi(e) == 4);
  }();
  return v;
}

int chunk_set_func_attributes() {
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_sweep),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_sweep8),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLds));
  return ACINO_OK;
}

int chunk_reduce(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                 const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                 Profiler* prof) {
  {
    ProfSpan span(prof, PC_CHUNK_SWEEP, s, pl.n_nodes - pl.n_sep);
    if (sweep_eight_waves())
      hipLaunchKernelGGL(k_chunk_sweep8, dim3(pl.n_chunks), dim3(SW8_T), kSweepLds, s, ch, sp, d_c, d_numeric_err, d_status,
                         pl.m, pl.n_chunks);
    else
      hipLaunchKernelGGL(k_chunk_sweep, dim3(pl.n_chunks), dim3(256), kSweepLds, s, ch, sp, d_c, d_numeric_err, d_status, pl.m,
                         pl.n_chunks);
  }
  ACINO_LAUNCH_CHECK();
  if (pl.n_sep == 0) return ACINO_OK;
  {
    ProfSpan span(prof, PC_SEP_COMBINE, s, pl.n_sep);
    hipLaunchKernelGGL(k_sep_combine, dim3(pl.n_sep), dim3(256), 0, s, sp, d_status);
  }
  ACINO_LAUNCH_CHECK();
  return bcr_reduce(sepch, sepsch, d_c, d_numeric_err, d_status, s, prof);
}

int chunk_backsub(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                  const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                  Profiler* prof) {
  if (pl.n_sep > 0) {
    int rc = bcr_backsub(sepch, sepsch, d_c, d_status, s, prof, d_numeric_err);
    if (rc) return rc;
  }
  {
    ProfSpan span(prof, PC_CHUNK_BACKSUB, s, pl.n_nodes - pl.n_sep);
    hipLaunchKernelGGL(k_chunk_backsub, dim3(pl.n_chunks), dim3(256), 0, s, ch, sp, d_c, d_status, pl.m, pl.n_chunks);
  }
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

}  // namespace acino
