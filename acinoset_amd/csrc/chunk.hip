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

void ChunkPlan::build(int nodes_total, int chunk_nodes, bool pin_l, bool pin_r) {
  n_nodes = nodes_total;
  m = n_chunks = n_sep = 0;
  node0 = pin_l ? 1 : 0;
  pin_right = pin_r ? 1 : 0;
  const int nodes = nodes_total - node0;             // the swept part of the chain (a right pin is its last node)
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
  // a right pin must not be a run of its own (a run builds its right separator from at least one interior node): the last
  // run then takes m + 1 nodes
  if (pin_r && n_chunks > 1 && nodes - (n_chunks - 1) * mm == 1) --n_chunks;
  n_sep = n_chunks - 1 + node0 + pin_right;
}

// ---- tile helpers on LDS matrices (leading dimension LD): one 16x16 output tile per call, all operands read first ----
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

// LDS of the sweep kernel: three 80 x 81 matrices (U_k | D~_k+1 -> U_k+1 | spike) + tables = 160.5 KB, one workgroup per CU.
constexpr int SW_VEC = 18 * NP + BS + 8 + 8 + 3 * NP + 64;   // cL cR | bv | red | sync | kq klo khi | debug stamps
static constexpr size_t kSweepLds = (3 * MAT + SW_VEC) * sizeof(double);

// The value of x, made opaque to the optimiser: address arithmetic derived from it cannot be hoisted out of the node loop
// (hoisted, the hundreds of per-tile LDS / HBM addresses of this kernel end up spilled to scratch and are reloaded one by
// one in front of the loads that need them).
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
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

// ================================================================================================================
// The sweep kernel (eight waves).  Per node k of a run (D~_k, F_k, b~_k in LDS; see the head of the file) the loop has two dependency chains that only meet once per node:
//   D chain     : U_k -> G_k = U_k U_k^T -> D~_k+1 = D_k+1 - E^T G_k E -> Cholesky -> U_k+1            (Xc -> Xn)
//   spike chain : F_k -> W = U_k^T F_k -> D_L -= W^T W,  T_k = U_k W -> HBM,  F_k+1 = -E^T T_k          (Xc, Y)
// Per node: a SERIAL part on all eight waves (G_k, its store, the two stencil passes, the next node built), then a
// PARALLEL part in which waves 0..2 factor the next node (wave 0: the five 16-pivot chains with the look-ahead tile;
// waves 1, 2: panel and trailing tiles, alternate products) while waves 3..7 run the WHOLE spike chain of node k, each
// on its own 16-column strip of the spike (strip = wave - 3; only W^T W reads across strips: two 5-wave LDS-counter
// barriers per node).  Two waves share every SIMD, so the matrix-core work of the spike fills the issue slots the
// pivot chains leave empty.  The right-hand side rides along as COLUMN 79 of the spike (the left separator has 75 unknowns,
// columns 75..79 are free): column 79 of W is y = U^T b~, of T it is z = D~^-1 b~, of F_k+1 it is -E^T z (+ b_k+1), and row 79
// of W^T W is y^T W = the left separator's right-hand-side update - no mat-vec phases.
constexpr int SW_T = 512;
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
// busy (helpers of the two-team sweep only, else null): the word a helper keeps at 1 while it has work - lowered in front of every
// wait, raised behind it - for the strip wave that shares its SIMD to yield to
__device__ __forceinline__ void helper_wait_begin(int* busy, int lane) {
  if (busy && lane == 0) __hip_atomic_store(busy, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void helper_wait_end(int* busy, int lane) {
  if (busy && lane == 0) __hip_atomic_store(busy, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void chol80_trio(double* Lm, int role, int lane, int* err, int* sync, int& t3, int& t2, int& posted,
                                            long long* dbg = nullptr, int* busy = nullptr) {
  const int li = lane & 15, lk = lane >> 4;
  if (role == 0) {
    __builtin_amdgcn_s_setprio(3);                     // the chain's VALU wins the issue arbitration against its SIMD mate
    chol16_inv(Lm, lane, err);
  } else {
    __builtin_amdgcn_s_setprio(2);                     // a helper's short bursts go ahead of its SIMD mate's strip products
  }
  helper_wait_begin(busy, lane);
  sub_barrier(sync, t3, 3, lane);
  helper_wait_end(busy, lane);
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
      __builtin_amdgcn_s_setprio(0);
      helper_wait_begin(busy, lane);
      sub_barrier(sync, t3, 3, lane);
      helper_wait_end(busy, lane);
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
      helper_wait_begin(busy, lane);
      sub_barrier(sync + 2, t2, 2, lane);              // the helpers' three panel tiles
      while (__hip_atomic_load(sync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < posted) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      helper_wait_end(busy, lane);
      if (dbg && lane == 0 && role == 1) dbg[56 + kb] = (long long)wall_clock64();
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
    helper_wait_begin(busy, lane);
    sub_barrier(sync, t3, 3, lane);
    helper_wait_end(busy, lane);
  }
}

// ---- the three-wave Cholesky, round-5 protocol -------------------------------------------------------------------------------
// chol80_trio meets in a three-wave barrier after every block column: the pivot-chain wave waits for ALL trailing products of
// step kb before it may touch block column kb + 1, so whatever slows a helper (the strip wave it shares a SIMD with: matrix
// instructions of the younger wave slip into every stall of the older one, 64 cycles each) lands on the chain.  But the chain
// needs only TWO of those products - (kb+2, kb+1), its next panel tile, and (kb+2, kb+2), its next look-ahead tile, the first
// entries of the step's list, one per helper - and nothing else the helpers write until the last block column.  Here the
// helpers signal those two ("crit") and the chain waits for nothing else; the helpers follow the chain's posts (U_kk stored,
// panel tile stored) and meet each other at the end of a step.  They may fall a block column behind without the chain noticing.
//   sync[12]: crit (2 per step kb = 0, 1, 2)   sync[13]: the chain's posts (U_00, then panel kb / U_kb+1,kb+1 for kb = 0 .. 3)
struct TrioSync {
  int t3 = 0, t2 = 0, posts = 0, crit = 0;             // rounds / counts so far (the counters are never reset)
};
template <int KB, int PART, int NPARTS, int Q0, int NQ>
__device__ __forceinline__ void trail_group(double* Lm, const double (&P)[NT][4], int li, int lk) {
  constexpr TrailList<KB> TL{};
  d4 a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ti = TL.ti[PART + NPARTS * (Q0 + q)], tj = TL.tj[PART + NPARTS * (Q0 + q)];
    const double* Cc = Lm + (ti * 16) * LD + tj * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) a[q][rr] = ti == KB ? 0.0 : Cc[(lk + 4 * rr) * LD + li];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ti = TL.ti[PART + NPARTS * (Q0 + q)], tj = TL.tj[PART + NPARTS * (Q0 + q)];
      a[q] = mfma(-P[ti][s], P[tj][s], a[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ti = TL.ti[PART + NPARTS * (Q0 + q)], tj = TL.tj[PART + NPARTS * (Q0 + q)];
    double* Cc = Lm + (ti * 16) * LD + tj * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = a[q][rr];
  }
}
// the step's products of one helper; CRIT: the first one on its own, stored and signalled before the others start
template <int KB, int PART, bool CRIT>
__device__ __forceinline__ void trail_step2(double* Lm, int li, int lk, int* ccrit, int lane) {
  double P[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) P[t][s] = Lm[(t * 16 + li) * LD + KB * 16 + 4 * s + lk];
  constexpr TrailList<KB> TL{};
  constexpr int NTOT = (TL.n - PART + 1) / 2;
  constexpr int F = CRIT ? 1 : 0;
  if constexpr (CRIT) {
    trail_group<KB, PART, 2, 0, 1>(Lm, P, li, lk);
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ccrit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  }
  if constexpr (NTOT - F >= 1) trail_group<KB, PART, 2, F, (NTOT - F < 4 ? NTOT - F : 4)>(Lm, P, li, lk);
  if constexpr (NTOT - F > 4) trail_group<KB, PART, 2, F + 4, NTOT - F - 4>(Lm, P, li, lk);
}
__device__ __forceinline__ void spin_until(int* f, int target) {
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void post_one(int* f, int lane) {
  asm volatile("" ::: "memory");                       // (LDS performs a wave's operations in issue order: the stores in front are visible first)
  if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void bar_n(int* cnt, int& target, int n, int lane) {
  target += n;
  post_one(cnt, lane);
  spin_until(cnt, target);
}
__device__ __forceinline__ void chol80_trio2(double* Lm, int role, int lane, int* err, int* sync, TrioSync& ts,
                                             long long* dbg = nullptr) {
  const int li = lane & 15, lk = lane >> 4;
  int* const c3 = sync;
  int* const c2 = sync + 2;
  int* const ccrit = sync + 12;
  int* const cpost = sync + 13;
  const int post0 = ts.posts, crit0 = ts.crit;
  if (role == 0) {
    __builtin_amdgcn_s_setprio(3);                     // the chain's VALU wins the issue arbitration against its SIMD mate
    chol16_inv(Lm, lane, err);
    post_one(cpost, lane);                             // U_00
  } else {
    __builtin_amdgcn_s_setprio(2);
  }
#pragma unroll 1
  for (int kb = 0; kb < NT; ++kb) {
    if (kb == NT - 1) {
      bar_n(c3, ts.t3, 3, lane);                       // U_44 stored, every trailing product of step 3 stored
    } else if (role == 0) {
      if (kb > 0) spin_until(ccrit, crit0 + 2 * kb);   // tiles (kb+1, kb) and (kb+1, kb+1) carry the update of step kb - 1
    } else {
      spin_until(cpost, post0 + 2 * kb + 1);           // U_kk
    }
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
      __builtin_amdgcn_s_setprio(0);
      bar_n(c3, ts.t3, 3, lane);
      break;
    }
    if (role == 0) {
      post_one(cpost, lane);                           // panel tile (kb+1, kb)
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
      post_one(cpost, lane);                           // U_kb+1,kb+1
      if (dbg && lane == 0) dbg[41 + 2 * kb] = (long long)wall_clock64();
    } else {
      bar_n(c2, ts.t2, 2, lane);                       // the helpers' three panel tiles
      spin_until(cpost, post0 + 2 * kb + 2);           // the chain's panel tile
      if (dbg && lane == 0 && role == 1) dbg[56 + kb] = (long long)wall_clock64();
      if (role == 1) {
        if (kb == 0) trail_step2<0, 0, true>(Lm, li, lk, ccrit, lane);
        else if (kb == 1) trail_step2<1, 0, true>(Lm, li, lk, ccrit, lane);
        else if (kb == 2) trail_step2<2, 0, true>(Lm, li, lk, ccrit, lane);
        else trail_step2<3, 0, false>(Lm, li, lk, ccrit, lane);
      } else {
        if (kb == 0) trail_step2<0, 1, true>(Lm, li, lk, ccrit, lane);
        else if (kb == 1) trail_step2<1, 1, true>(Lm, li, lk, ccrit, lane);
        else if (kb == 2) trail_step2<2, 1, true>(Lm, li, lk, ccrit, lane);
        else trail_step2<3, 1, false>(Lm, li, lk, ccrit, lane);
      }
      if (dbg && lane == 0) dbg[48 + 4 * (role - 1) + kb] = (long long)wall_clock64();
      bar_n(c2, ts.t2, 2, lane);                       // every product of the step stored: the next panel column is complete
    }
  }
  ts.posts = post0 + 9;
  ts.crit = crit0 + 6;
}

// ---- the three-wave Cholesky, split by what the pivot chain needs ----------------------------------------------------------------
// chol80_trio2 lets the helpers fall behind, but both still carry half of EVERYTHING, and everything includes the 30 tile
// products that only build U = L^-T (needed when the factorisation is over, for G): a step's worth of work per helper stays
// longer than the chain's step, the lag grows and the chain ends up waiting all the same.  Here the helpers are split by
// deadline:
//   helper L (role 1, the OLDER wave of its SIMD: it wins the issue arbitration): the panel tiles below the chain's and the
//     LOWER trailing products - 12 / 7 / 3 / 0 tile products at block column 0 / 1 / 2 / 3 against the chain's ~2 us per column;
//     the chain's two tiles first, signalled (crit);
//   helper U (role 2): the panel and trailing products of the strictly-upper tiles (U), 4 / 7 / 8 / 7 products, in the gaps the
//     first leaves on their common matrix pipe; it follows the chain's posts and helper L's "panel of column kb stored" and
//     has no deadline before the last block column.
// Last block column: the four tiles (t, 4) U_44 wait for U_44 and for helper U's last products, then one barrier of the three.
//   sync[12]: crit (1 per block column 0 .. 2)   [13]: chain posts   [14]: helper L's panel columns   [15]: helper U done with step 3
// (block column 0 is the long one - 3 panel tiles + 9 lower products against helper U's 4 - and sets the distance the helper
//  keeps to the chain for the rest of the factorisation: there row 4 of the lower products goes to helper U, who has them stored
//  and signalled (culow) before helper L touches row 4 of block column 1)
template <int KB, int R0 = KB + 2, int R1 = NT - 1>
struct LowerList {                                     // (r, c), r = R0 .. R1, c = KB+1 .. r: row KB+2 first (the chain's tiles)
  int ti[9], tj[9], n;
  constexpr LowerList() : ti{}, tj{}, n(0) {
    for (int r = R0; r <= R1; ++r)
      for (int c = KB + 1; c <= r; ++c) {
        ti[n] = r;
        tj[n] = c;
        ++n;
      }
  }
};
template <int KB>
struct UpperList {                                     // (r, c), r = 0 .. KB, c = KB+1 .. 4: tiles of U, first written at r == KB
  int ti[16], tj[16], n;
  constexpr UpperList() : ti{}, tj{}, n(0) {
    for (int r = 0; r <= KB; ++r)
      for (int c = KB + 1; c < NT; ++c) {
        ti[n] = r;
        tj[n] = c;
        ++n;
      }
  }
};
// products Q0 .. Q0 + NQ - 1 of a list: C(ti, tj) -= P(ti) P(tj)^T, P = the panel tiles of block column KB (registers)
template <class LIST, int KB, int Q0, int NQ>
__device__ __forceinline__ void trail_run(double* Lm, const double (&P)[NT][4], int li, int lk) {
  constexpr LIST TL{};
  d4 a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ti = TL.ti[Q0 + q], tj = TL.tj[Q0 + q];
    const double* Cc = Lm + (ti * 16) * LD + tj * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) a[q][rr] = ti == KB ? 0.0 : Cc[(lk + 4 * rr) * LD + li];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = mfma(-P[TL.ti[Q0 + q]][s], P[TL.tj[Q0 + q]][s], a[q]);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double* Cc = Lm + (TL.ti[Q0 + q] * 16) * LD + TL.tj[Q0 + q] * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = a[q][rr];
  }
}
template <class LIST, int KB, int Q0>
__device__ __forceinline__ void trail_rest(double* Lm, const double (&P)[NT][4], int li, int lk) {
  constexpr LIST TL{};
  if constexpr (Q0 < TL.n) {
    constexpr int NQ = TL.n - Q0 < 4 ? TL.n - Q0 : 4;
    trail_run<LIST, KB, Q0, NQ>(Lm, P, li, lk);
    trail_rest<LIST, KB, Q0 + NQ>(Lm, P, li, lk);
  }
}
template <int KB>
__device__ __forceinline__ void load_panel(double (&P)[NT][4], const double* Lm, int li, int lk) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) P[t][s] = Lm[(t * 16 + li) * LD + KB * 16 + 4 * s + lk];
}
// helper L, block column KB: the lower trailing products; the first two (the chain's next panel and look-ahead tiles) signalled.
// Every operand of the step - the panel column and all accumulator tiles - is requested before the first product: one LDS round trip.
template <int KB, class LIST = LowerList<KB>>
__device__ __forceinline__ void helperL_trailing(double* Lm, int li, int lk, int* ccrit, int lane) {
  constexpr LIST TL{};
  double P[NT][4];
  d4 a[TL.n];
  load_panel<KB>(P, Lm, li, lk);
#pragma unroll
  for (int q = 0; q < TL.n; ++q) {
    const double* Cc = Lm + (TL.ti[q] * 16) * LD + TL.tj[q] * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) a[q][rr] = Cc[(lk + 4 * rr) * LD + li];
  }
  constexpr int NC = TL.n < 2 ? TL.n : 2;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < NC; ++q) a[q] = mfma(-P[TL.ti[q]][s], P[TL.tj[q]][s], a[q]);
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    double* Cc = Lm + (TL.ti[q] * 16) * LD + TL.tj[q] * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = a[q][rr];
  }
  post_one(ccrit, lane);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = NC; q < TL.n; ++q) a[q] = mfma(-P[TL.ti[q]][s], P[TL.tj[q]][s], a[q]);
#pragma unroll
  for (int q = NC; q < TL.n; ++q) {
    double* Cc = Lm + (TL.ti[q] * 16) * LD + TL.tj[q] * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = a[q][rr];
  }
}
template <int KB>
__device__ __forceinline__ void helperU_trailing(double* Lm, int li, int lk) {
  double P[NT][4];
  load_panel<KB>(P, Lm, li, lk);
  trail_rest<UpperList<KB>, KB, 0>(Lm, P, li, lk);
}
// NQ panel tiles t0 .. t0 + NQ - 1 of block column kb: tile(t, kb) <- tile(t, kb) U_kk
template <int NQ>
__device__ __forceinline__ void panel_tiles(double* Lm, int kb, int t0, int li, int lk) {
  const double* Ukk = Lm + (kb * 16) * LD + kb * 16;
  double bq[4], av[NQ][4];
  d4 acc[NQ];
#pragma unroll
  for (int s = 0; s < 4; ++s) bq[s] = Ukk[(4 * s + lk) * LD + li];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double* A = Lm + ((t0 + q) * 16) * LD + kb * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) av[q][s] = A[li * LD + 4 * s + lk];
    acc[q] = d4{0, 0, 0, 0};
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = mfma(av[q][s], bq[s], acc[q]);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double* A = Lm + ((t0 + q) * 16) * LD + kb * 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) A[(lk + 4 * rr) * LD + li] = acc[q][rr];
  }
}
struct Trio3Sync {
  int t3 = 0, posts = 0, crit = 0, lpan = 0, udone = 0, ulow = 0;
};
__device__ __forceinline__ void chol80_trio3(double* Lm, int role, int lane, int* err, int* sync, Trio3Sync& ts,
                                             long long* dbg = nullptr) {
  const int li = lane & 15, lk = lane >> 4;
  int* const c3 = sync;
  int* const ccrit = sync + 12;
  int* const cpost = sync + 13;
  int* const clpan = sync + 14;
  int* const cudone = sync + 15;
  int* const culow = sync + 11;                        // helper U: row 4 of block column 0's lower products stored
  const int post0 = ts.posts, crit0 = ts.crit, lp0 = ts.lpan;
  if (role == 0) {
    // ---------------- the pivot chains ----------------
    __builtin_amdgcn_s_setprio(3);
    chol16_inv(Lm, lane, err);
    post_one(cpost, lane);                             // U_00
#pragma unroll 1
    for (int kb = 0; kb < NT - 1; ++kb) {
      if (kb > 0) spin_until(ccrit, crit0 + kb);       // tiles (kb+1, kb) and (kb+1, kb+1) carry the update of step kb - 1
      panel_tiles<1>(Lm, kb, kb + 1, li, lk);
      post_one(cpost, lane);                           // panel tile (kb+1, kb)
      if (dbg && lane == 0) dbg[40 + 2 * kb] = (long long)wall_clock64();
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
      post_one(cpost, lane);                           // U_kb+1,kb+1
      if (dbg && lane == 0) dbg[41 + 2 * kb] = (long long)wall_clock64();
    }
    spin_until(cudone, ts.udone + 1);                  // every tile of U carries the update of step 3
    panel_tiles<1>(Lm, NT - 1, 0, li, lk);             // (0, 4) U_44
    __builtin_amdgcn_s_setprio(0);
  } else if (role == 1) {
    // ---------------- helper L ----------------
    __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
    for (int kb = 0; kb < NT - 1; ++kb) {
      spin_until(cpost, post0 + 2 * kb + 1);           // U_kk
      if (kb == 1) spin_until(culow, ts.ulow + 1);     // row 4 carries the update of step 0
      if (kb == 0) panel_tiles<3>(Lm, 0, 2, li, lk);
      else if (kb == 1) panel_tiles<2>(Lm, 1, 3, li, lk);
      else if (kb == 2) panel_tiles<1>(Lm, 2, 4, li, lk);
      post_one(clpan, lane);                           // panel tiles (kb+2 .., kb) stored
      spin_until(cpost, post0 + 2 * kb + 2);           // the chain's panel tile (kb+1, kb)
      if (dbg && lane == 0) dbg[56 + kb] = (long long)wall_clock64();
      if (kb == 0) helperL_trailing<0, LowerList<0, 2, 3>>(Lm, li, lk, ccrit, lane);
      else if (kb == 1) helperL_trailing<1>(Lm, li, lk, ccrit, lane);
      else if (kb == 2) helperL_trailing<2>(Lm, li, lk, ccrit, lane);
      else post_one(ccrit, lane);
      if (dbg && lane == 0) dbg[48 + kb] = (long long)wall_clock64();
    }
    spin_until(cpost, post0 + 2 * (NT - 1) + 1);       // U_44
    spin_until(cudone, ts.udone + 1);
    panel_tiles<2>(Lm, NT - 1, 1, li, lk);             // (1, 4), (2, 4)
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ---------------- helper U ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int kb = 0; kb < NT - 1; ++kb) {
      spin_until(cpost, post0 + 2 * kb + 1);           // U_kk
      if (kb == 1) panel_tiles<1>(Lm, 1, 0, li, lk);
      else if (kb == 2) panel_tiles<2>(Lm, 2, 0, li, lk);
      else if (kb == 3) panel_tiles<3>(Lm, 3, 0, li, lk);
      spin_until(cpost, post0 + 2 * kb + 2);           // the chain's panel tile
      spin_until(clpan, lp0 + kb + 1);                 // helper L's panel tiles
      if (kb == 0) {
        double P[NT][4];
        load_panel<0>(P, Lm, li, lk);
        trail_rest<LowerList<0, 4, 4>, 0, 0>(Lm, P, li, lk);     // (4, 1) .. (4, 4)
        post_one(culow, lane);
        trail_rest<UpperList<0>, 0, 0>(Lm, P, li, lk);
      } else if (kb == 1) helperU_trailing<1>(Lm, li, lk);
      else if (kb == 2) helperU_trailing<2>(Lm, li, lk);
      else helperU_trailing<3>(Lm, li, lk);
      if (dbg && lane == 0) dbg[52 + kb] = (long long)wall_clock64();
    }
    post_one(cudone, lane);
    spin_until(cpost, post0 + 2 * (NT - 1) + 1);       // U_44
    panel_tiles<1>(Lm, NT - 1, 3, li, lk);             // (3, 4)
    __builtin_amdgcn_s_setprio(0);
  }
  bar_n(c3, ts.t3, 3, lane);
  ts.posts = post0 + 9;
  ts.crit = crit0 + 4;
  ts.lpan = lp0 + 4;
  ts.udone += 1;
  ts.ulow += 1;
}

// The NQ tiles (ib[q], jb[q]) of the left separator's update D_L -= W^T W that one spike wave owns: accumulators
// loaded from / stored to the workgroup's block in global memory (L2-resident between nodes).
template <int NQ>
__device__ __forceinline__ void syrk_load_n(d4 (&acc)[NQ], const double* __restrict__ Ag, bool load, const int (&ib)[NQ],
                                            const int (&jb)[NQ], int li, int lk) {
  // (loads unconditional; the select - syrk_mask_n - where the values are first needed, behind the W strip: a conditional
  //  load compiles to a branch per element with a full wait in front, and a select right behind the loads made the wave
  //  wait for them - three serialised round trips - before its W strip instead of after it)
  (void)load;
  const double* base = Ag + lk * BS + li;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[q][rr] = base[(ib[q] * 16 + 4 * rr) * BS + jb[q] * 16];
}
template <int NQ>
__device__ __forceinline__ void syrk_mask_n(d4 (&acc)[NQ], bool load) {
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

__global__ void __launch_bounds__(SW_T)
k_chunk_sweep(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, int* numeric_err, const int* __restrict__ status,
               int m, int n_chunks, int node0, int pin_right) {
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
  const int first = node0 + c * m;
  const bool hasL = c > 0 || node0 > 0, hasR = c + 1 < n_chunks || pin_right;
  const int len = c + 1 < n_chunks ? m : ch.n_nodes - first;       // (the last run: whatever is left, m + 1 at most)
  const int n_int = hasR ? len - 1 : len;
  const int sL = c - 1 + node0, sR = c + node0;                    // separator-chain indices of the run's two ends
  const size_t MB = (size_t)BS * BS;
  const int role = __builtin_amdgcn_readfirstlane(role8(wave));
  const bool uni_tables = coupling_tables_uniform(K, first, first + n_int - 1);   // (then the tables of `first` serve every node)
  const int n_spike = hasL ? 5 : 1;          // first run: only the right-hand side column (strip 4) is alive
  int t3 = 0, t2 = 0, tflag = 0, t5 = 0;     // rounds of the wave-subset barriers
  // max |projected gradient| over the rows this thread builds, over the whole run: ONE entry of gn_part per run, published
  // at the end (it was one per node: a shuffle tree and a barrier in every node's serial part, 14 x as many entries for
  // k_totals to walk)
  double gmax_run = 0.0;
  // (debug stamps: workgroup dbg[64] writes wall-clock ticks of its phases at node dbg[65] into dbg[0..63]; the selectors
  //  sit OUTSIDE the stamp range and are read ONCE)
  long long* const dbgp = (ch.dbg && (long long)blockIdx.x == ch.dbg[64]) ? ch.dbg : nullptr;
  const int dbg_k = dbgp ? (int)ch.dbg[65] : -1;
  // (stamps go to LDS: a global store in front of a release fence would itself delay the wave that is being timed)
#define SW_STAMP(i) do { if (dbgp && k == dbg_k && lane == 0) lst[i] = (long long)wall_clock64(); } while (0)
  if (tid < 4) sync[tid] = 0;
  if (tid < 64) lst[tid] = 0;
  if (c == 0 && node0) {                               // the left pin owns no frames here: its block starts from zero, the
    for (int e = tid; e < BS * BS; e += SW_T) sp.D[e] = 0.0;       // run's spike contribution AL is added by the reduction
    if (tid < BS) sp.b[tid] = 0.0;
  }
  if (tid < NP) {
    kq[tid] = K.q_w[tid];
    klo[tid] = K.lo[tid];
    khi[tid] = K.hi[tid];
  }
  __syncthreads();

  {  // ---- first node of the run, its spike F_0 = E_l (dense form) with the right-hand side in column 79
    NodeFetch f;
    build_fetch<SW_T>(f, ch, K, first, tid);
    fill_coupling_coef<SW_T>(cL, cR, K, first, tid, kq);
    for (int e = tid; e < MAT; e += SW_T) Y[e] = 0.0;
    gmax_run = build_finish<SW_T>(Xc, bv, f, K, first, tid, kq, klo, khi);
    __syncthreads();                                   // node, bv, tables, zeros complete
    if (hasL)
      for (int e = tid; e < 9 * NP; e += SW_T) {
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
    const int fb_next = 3 * (next - node0);            // (node t holds the local frames 3 (t - pin_left) ..)
    double hq[2][3] = {{0, 0, 0}, {0, 0, 0}}, xq[3] = {0, 0, 0}, gq[3] = {0, 0, 0}, cvq[3] = {0, 0, 0}, lamq = 0.0;
    bool liveq[3] = {false, false, false}, ownq[3] = {false, false, false};
    const int e1 = tid_ + SW_T;
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
      if (k > 0 && !uni_tables) fill_coupling_coef<SW_T>(cL, cR, K, node, tid_, kq);   // (beside the matrix-core work above)
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
      // G_k -> HBM, lower tiles only (G is symmetric: the back-substitution mirrors them)
      double* Gg = ch.D + node * MB;
#pragma unroll
      for (int q = 0; q < (LOWER_ITEMS + SW_T - 1) / SW_T; ++q) {
        const int idx = tid_ + SW_T * q;
        if (idx < LOWER_ITEMS) {
          int r, cc;
          lower_item(idx, r, cc);
          *reinterpret_cast<double2*>(Gg + r * BS + cc) = make_double2(Xn[r * LD + cc], Xn[r * LD + cc + 1]);
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
                if (ownq[j]) gmax_run = fmax(gmax_run, fabs(bb));   // (window sharding: owned frames only)
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
      // (padding rows / columns 75 .. 79 need no write: the identity block of D~_k gives an identity block of G_k = D~_k^-1,
      //  exactly - every product behind it has an exact 0 or 1 factor -, and G_k is what Xn holds there; bv[75 .. 79] = 0
      //  since the first node was built)
      if (wave == 0) SW_STAMP(3);
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
      double* Ag = sp.AL + (size_t)(hasL ? opaque(sL) : 0) * MB;   // (global address space kept: no pointer through asm)
      // the left separator's update so far (HBM / L2, owned by this workgroup): requested now, needed after W
      // (kept in memory between nodes, not in registers: measured in round 4 - the register form runs the sweep in 303.0 us
      //  against 303.7 us, the load is hidden behind the W strip - and memory does not depend on how the register allocator
      //  treats values that live across the lane-masked regions of the serial part; NOTES_perf.md)
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
      if (hasL) {
        syrk_mask_n<3>(accL, k > 0);
        syrk_run_n<3>(accL, Y, Ag, sib, sjb, li, lk);
      }
      SW_STAMP(18 + 4 * sw);
      d4 town[NT];
      strip_product<false>(Xc, Y, sw, town, li, lk);   // T strip = U W strip (reads its own strip of W only)
      sub_barrier(sync + 1, t5, n_spike, ln);          // nobody reads W any more
#pragma unroll
      for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, sw, town[ib], li, lk);
      SW_STAMP(19 + 4 * sw);
      // ---- z_k (column 79 of T) -> HBM, then F_k+1 = -E^T T_k in place, column 79 += the next node's right-hand side.
      //      Own strip only: no barrier (LDS operations of a wave are ordered).  T_k itself is not stored: the
      //      back-substitution regenerates T_k x_L from G_k.
      const int c_lo = 16 * sw;
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
        double2* d2 = reinterpret_cast<double2*>(sp.D + (size_t)sR * MB);
        for (int idx = tid; idx < BS * BS / 2; idx += SW_T) {
          const int e = 2 * idx, r = e / BS, cc = e % BS;
          d2[idx] = make_double2(Xn[r * LD + cc], Xn[r * LD + cc + 1]);
        }
      }
      if (tid < BS) sp.b[(size_t)sR * BS + tid] = Y[tid * LD + (BS - 1)];
      if (hasL) {
        double* Cg = sp.Cpl + (size_t)sL * MB;         // block(R, L): rows R, columns L
        for (int e = tid; e < BS * BS; e += SW_T) {
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
  publish_gmax<SW_T>(gmax_run, red, ch.gn_part, c, tid);
}

// ================================================================================================================
// The sweep kernel, round-5 form.  Same elimination, same outputs; what changed is the SPIKE algebra and the buffer roles:
//   T_k = G_k F_k               (G_k = D~_k^-1 is formed anyway for the next node: the spike no longer goes through
//   D_L -= F_k^T T_k             W = U^T F, W^T W, T = U W - 160 matrix instructions per strip instead of 180, ONE barrier of
//   F_k+1 = -E^T T_k             the strip waves per node instead of two, no W written back and read again)
// A strip wave keeps its strip of F in registers (the accumulator layout of a 16 x 16 tile IS the B-operand layout of the
// four k-steps over it), reads G - symmetric, stored in full - from LDS, and keeps the finished T strip in registers as
// the B operand of its three tiles of F^T T (tile pairs {j, j}, {j, j + 1}, {j, j + 2} mod 5: every unordered pair of strips
// once, 60 instructions per wave; the other operand is a strip of F, read-only in Y until the one barrier).
// Buffers: Xf = the factorisation's workspace (D~_k -> [L \ U_k]; after G_k is formed U_k is dead and D~_k+1 is built
// straight into it), Xg = G_k for the whole node (stencil pass, spike, and the store to HBM - by the spike waves, off the
// factor chain's path), Y = the spike.  No buffer swap, one barrier fewer in the serial part (the stencil reads Xg and
// writes Xf).
__device__ __forceinline__ void spike_gf(const double* G, const double (&bF)[NT][4], d4 (&acc)[NT], int li, int lk, int mode = 0) {
  constexpr int NSTEP = 4 * NT;
  double a[3][NT];
  const double* gb = G + lk * LD + li;                 // A(ib, kb)[li][4 s + lk] = G[kb 16 + 4 s + lk][ib 16 + li] (symmetric)
  auto fetch = [&](int buf, int step) {
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) a[buf][ib] = mode == 1 ? 1e-3 * (li + step) : gb[(4 * step) * LD + ib * 16];
  };
#pragma unroll
  for (int ib = 0; ib < NT; ++ib) acc[ib] = d4{0, 0, 0, 0};
  fetch(0, 0);
  fetch(1, 1);
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    if (step + 2 < NSTEP) fetch((step + 2) % 3, step + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) {
      if (mode == 2) acc[ib][0] += a[step % 3][ib];
      else acc[ib] = mfma(a[step % 3][ib], bF[step >> 2][step & 3], acc[ib]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// acc[q] -= F(:, as[q])^T T(:, own strip): A operand = the strip as[q] of F in Y (column pattern), B = the T tiles in registers
template <int NQ>
__device__ __forceinline__ void spike_ftt(d4 (&acc)[NQ], const double* Yf, const d4 (&T)[NT], const int (&as)[NQ], int li, int lk, int mode = 0) {
  constexpr int NSTEP = 4 * NT;
  const double* p = Yf + lk * LD + li;
  double a0[NQ], a1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) a0[q] = mode == 1 ? 1e-3 * li : p[as[q] * 16];
#pragma unroll
  for (int st = 0; st < NSTEP; st += 2) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) a1[q] = mode == 1 ? 1e-3 * (li + st) : p[(4 * (st + 1)) * LD + as[q] * 16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { if (mode == 2) acc[q][0] += a0[q]; else acc[q] = mfma(-a0[q], T[st >> 2][st & 3], acc[q]); }
    __builtin_amdgcn_sched_barrier(0);
    if (st + 2 < NSTEP) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) a0[q] = mode == 1 ? 1e-3 * (li - st) : p[(4 * (st + 2)) * LD + as[q] * 16];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { if (mode == 2) acc[q][0] += a1[q]; else acc[q] = mfma(-a1[q], T[(st + 1) >> 2][(st + 1) & 3], acc[q]); }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ void __launch_bounds__(SW_T)
k_chunk_sweep2(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, int* numeric_err, const int* __restrict__ status,
                int m, int n_chunks, int node0, int pin_right) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* const Xf = reinterpret_cast<double*>(smem_raw);   // D~_k -> [L \ U_k] -> D~_k+1
  double* const Xg = Xf + MAT;                               // G_k
  double* const Y = Xg + MAT;                                // spike: F_k -> T_k -> F_k+1; column 79 = right-hand side
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
  const int first = node0 + c * m;
  const bool hasL = c > 0 || node0 > 0, hasR = c + 1 < n_chunks || pin_right;
  const int len = c + 1 < n_chunks ? m : ch.n_nodes - first;       // (the last run: whatever is left, m + 1 at most)
  const int n_int = hasR ? len - 1 : len;
  const int sL = c - 1 + node0, sR = c + node0;                    // separator-chain indices of the run's two ends
  const size_t MB = (size_t)BS * BS;
  const int role = __builtin_amdgcn_readfirstlane(role8(wave));
  const bool uni_tables = coupling_tables_uniform(K, first, first + n_int - 1);   // (then the tables of `first` serve every node)
  const int n_spike = hasL ? 5 : 1;          // first run: only the right-hand side column (strip 4) is alive
  int t3 = 0, t2 = 0, tflag = 0, t5 = 0;     // rounds of the wave-subset barriers
  double gmax_run = 0.0;
  long long* const dbgp = (ch.dbg && (long long)blockIdx.x == ch.dbg[64]) ? ch.dbg : nullptr;
  const int dbg_k = dbgp ? (int)ch.dbg[65] : -1;
  // (experiment knobs of the stamped workgroup: dbg[66] = mask of spike roles that skip their matrix-core work; dbg[67] <- SIMD
  //  index of every wave, four bits each, from HW_ID)
  const int dbg_skip = dbgp ? (int)ch.dbg[66] : 0;
  if (dbgp && lane == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    atomicOr(reinterpret_cast<unsigned long long*>(ch.dbg + 67), (unsigned long long)((hw >> 4) & 3) << (4 * wave));
    if (wave == 0) ch.dbg[68] = hw;
  }
#define SW_STAMP(i) do { if (dbgp && k == dbg_k && lane == 0) lst[i] = (long long)wall_clock64(); } while (0)
  if (tid < 4) sync[tid] = 0;
  if (tid < 64) lst[tid] = 0;
  if (c == 0 && node0) {                               // the left pin owns no frames here: its block starts from zero, the
    for (int e = tid; e < BS * BS; e += SW_T) sp.D[e] = 0.0;       // run's spike contribution AL is added by the reduction
    if (tid < BS) sp.b[tid] = 0.0;
  }
  if (tid < NP) {
    kq[tid] = K.q_w[tid];
    klo[tid] = K.lo[tid];
    khi[tid] = K.hi[tid];
  }
  __syncthreads();

  {  // ---- first node of the run, its spike F_0 = E_l (dense form) with the right-hand side in column 79
    NodeFetch f;
    build_fetch<SW_T>(f, ch, K, first, tid);
    fill_coupling_coef<SW_T>(cL, cR, K, first, tid, kq);
    for (int e = tid; e < MAT; e += SW_T) Y[e] = 0.0;
    gmax_run = build_finish<SW_T>(Xf, bv, f, K, first, tid, kq, klo, khi);
    __syncthreads();                                   // node, bv, tables, zeros complete
    if (hasL)
      for (int e = tid; e < 9 * NP; e += SW_T) {
        const int pair = e / NP, p = e % NP, ii = pair / 3, jj = pair % 3;
        if (ii <= jj) Y[(ii * NP + p) * LD + jj * NP + p] = cL[e];
      }
    if (tid < BS) Y[tid * LD + (BS - 1)] = bv[tid];
    __syncthreads();
    if (role < 3) chol80_trio(Xf, role, lane, numeric_err, sync, t3, t2, tflag);
    __syncthreads();
  }

#pragma unroll 1
  for (int k = 0; k < n_int; ++k) {
    const int node = first + k, next = node + 1;
    const bool last = k + 1 == n_int;
    const bool has_next = !last || hasR;
    // ================= serial part (all eight waves): G_k, then the next node =================
    const int tid_ = opaque(tid);
    const int fb_next = 3 * (next - node0);            // (node t holds the local frames 3 (t - pin_left) ..)
    double hq[2][3] = {{0, 0, 0}, {0, 0, 0}}, xq[3] = {0, 0, 0}, gq[3] = {0, 0, 0}, cvq[3] = {0, 0, 0}, lamq = 0.0;
    bool liveq[3] = {false, false, false}, ownq[3] = {false, false, false};
    const int e1 = tid_ + SW_T;
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
        if (t >= 0) g[q] = tile_u_ut(Xf, tri_i(t), tri_j(t), gi, gk);
      }
      asm volatile("" : "+v"(g[0][0]), "+v"(g[1][0]), "+v"(g[2][0]));
      SW_STAMP(32 + wave);
      if (k > 0 && !uni_tables) fill_coupling_coef<SW_T>(cL, cR, K, node, tid_, kq);   // (beside the matrix-core work above)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int t = gram8_tile(wave, q);
        if (t >= 0) {
          const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            Xg[(ib * 16 + gk + 4 * rr) * LD + jb * 16 + gi] = g[q][rr];
            if (ib != jb) Xg[(jb * 16 + gi) * LD + ib * 16 + gk + 4 * rr] = g[q][rr];
          }
        }
      }
    }
    __syncthreads();                                   // G in Xg, every read of U_k (Xf) done, tables of this node visible
    if (wave == 0) SW_STAMP(1);
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(hq[0][j]), "+v"(hq[1][j]), "+v"(xq[j]), "+v"(gq[j]));
    asm volatile("" : "+v"(lamq));
    if (wave == 0) SW_STAMP(2);
    if (has_next) {
      // S = E^T G_k E on the thread's blocks (reads Xg), the next node written straight into Xf
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int pr = sl ? pd1 : pd0, pc = sl ? pc1 : pc0;
        const bool dg = sl ? diag1 : diag0;
        if (sl == 0 || own1) {
          const double a00 = cR[0 * NP + pr], a01 = cR[1 * NP + pr], a02 = cR[2 * NP + pr];
          const double a11 = cR[4 * NP + pr], a12 = cR[5 * NP + pr], a22 = cR[8 * NP + pr];
          const double b00 = cR[0 * NP + pc], b01 = cR[1 * NP + pc], b02 = cR[2 * NP + pc];
          const double b11 = cR[4 * NP + pc], b12 = cR[5 * NP + pc], b22 = cR[8 * NP + pc];
          double mm[3][3], S[3][3];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            const double* gp = Xg + (jj * NP + pr) * LD + pc;
            const double g0 = gp[0], g1 = gp[NP], g2 = gp[2 * NP];
            mm[jj][0] = g0 * b00 + g1 * b01 + g2 * b02;
            mm[jj][1] = g1 * b11 + g2 * b12;
            mm[jj][2] = g2 * b22;
          }
#pragma unroll
          for (int ii = 0; ii < 3; ++ii) {
            S[0][ii] = a00 * mm[0][ii] + a01 * mm[1][ii] + a02 * mm[2][ii];
            S[1][ii] = a11 * mm[1][ii] + a12 * mm[2][ii];
            S[2][ii] = a22 * mm[2][ii];
          }
          double v[3][3];
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) v[j][jp] = (j == jp ? hq[sl][j] : 0.0);
          if (dg) {
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
                if (ownq[j]) gmax_run = fmax(gmax_run, fabs(bb));   // (window sharding: owned frames only)
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
            for (int jp = 0; jp < 3; ++jp) Xf[(j * NP + pr) * LD + jp * NP + pc] = v[j][jp] - S[j][jp];
        }
      }
      // (padding rows / columns 75 .. 79: the factorisation left an identity block there - L and U of an identity block are
      //  the identity, the blocks beside it exact zeros -; the diagonal is rewritten all the same: five stores)
      if (tid_ < BS - 3 * NP) Xf[(3 * NP + tid_) * LD + 3 * NP + tid_] = 1.0;
      if (wave == 0) SW_STAMP(3);
    }
    __syncthreads();
    if (wave == 0) SW_STAMP(4);
    // ================= parallel part =================
    if (role < 3) {
      if (!last)
        chol80_trio(Xf, role, opaque(lane), numeric_err, sync, t3, t2, tflag, (dbgp && k == dbg_k) ? lst : nullptr);
      SW_STAMP(8 + wave);
    } else {
      const int sw = __builtin_amdgcn_readfirstlane(opaque(role)) - 3;
      const int ln = opaque(lane);
      const int li = ln & 15, lk = ln >> 4;
      {
        // G_k -> HBM, lower tiles only (G is symmetric: the back-substitution mirrors them); 1920 items over the five waves
        double* Gg = ch.D + node * MB;
#pragma unroll
        for (int q = 0; q < LOWER_ITEMS / 320; ++q) {
          const int idx = sw * 64 + ln + 320 * q;
          int r, cc;
          lower_item(idx, r, cc);
          *reinterpret_cast<double2*>(Gg + r * BS + cc) = make_double2(Xg[r * LD + cc], Xg[r * LD + cc + 1]);
        }
      }
      if (hasL || sw == 4) {
        // the strip's three tiles of the left separator's update: pairs (a, sw), a = sw, sw + 1, sw + 2 mod 5; kept in the lower
        // tiles of AL (a < sw: stored transposed)
        int sa[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) sa[q] = (sw + q) % NT;
        double* Ag = sp.AL + (size_t)(hasL ? opaque(sL) : 0) * MB;
        auto al_at = [&](int q, int rr) -> double* {
          return sa[q] >= sw ? Ag + (size_t)(sa[q] * 16 + 4 * rr + lk) * BS + sw * 16 + li
                             : Ag + (size_t)(sw * 16 + li) * BS + sa[q] * 16 + 4 * rr + lk;
        };
        d4 accL[3] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
        if (hasL) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) accL[q][rr] = *al_at(q, rr);
        }
        d4 T[NT];
        {
          double bF[NT][4];
          const double* yb = Y + lk * LD + sw * 16 + li;
#pragma unroll
          for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) bF[kb][s] = yb[(kb * 16 + 4 * s) * LD];
          SW_STAMP(59 + sw);
          if (!((dbg_skip >> role) & 1)) spike_gf(Xg, bF, T, li, lk, (dbg_skip >> 8) & 3);
        }
        SW_STAMP(16 + 4 * sw);
        if (hasL) {
          syrk_mask_n<3>(accL, k > 0);
          if (!((dbg_skip >> role) & 1)) spike_ftt<3>(accL, Y, T, sa, li, lk, (dbg_skip >> 8) & 3);
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) *al_at(q, rr) = accL[q][rr];
        }
        SW_STAMP(17 + 4 * sw);
        sub_barrier(sync + 1, t5, n_spike, ln);          // nobody reads F_k any more
        SW_STAMP(18 + 4 * sw);
#pragma unroll
        for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, sw, T[ib], li, lk);
        SW_STAMP(19 + 4 * sw);
        const int c_lo = 16 * sw;
        if (sw == 4) {
          ch.b[(size_t)node * BS + ln] = Y[ln * LD + (BS - 1)];
          if (ln < 16) ch.b[(size_t)node * BS + 64 + ln] = Y[(64 + ln) * LD + (BS - 1)];
        }
        if (has_next) {
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
      }
      SW_STAMP(8 + wave);
    }
    __syncthreads();                                   // next node factored, F_k+1 complete, G_k no longer read
    if (wave == 0) SW_STAMP(7);
    if (dbgp && k == dbg_k) {
      __syncthreads();
      if (tid < 64 && lst[tid]) dbgp[tid] = lst[tid];
    }
    if (last && hasR) {                                // the node built last is the right separator
      {
        double2* d2 = reinterpret_cast<double2*>(sp.D + (size_t)sR * MB);
        for (int idx = tid; idx < BS * BS / 2; idx += SW_T) {
          const int e = 2 * idx, r = e / BS, cc = e % BS;
          d2[idx] = make_double2(Xf[r * LD + cc], Xf[r * LD + cc + 1]);
        }
      }
      if (tid < BS) sp.b[(size_t)sR * BS + tid] = Y[tid * LD + (BS - 1)];
      if (hasL) {
        double* Cg = sp.Cpl + (size_t)sL * MB;         // block(R, L): rows R, columns L
        for (int e = tid; e < BS * BS; e += SW_T) {
          const int r = e / BS, cc = e % BS;
          Cg[e] = cc < 3 * NP ? Y[r * LD + cc] : 0.0;
        }
      }
    }
  }
#undef SW_STAMP
  publish_gmax<SW_T>(gmax_run, red, ch.gn_part, c, tid);
}

// ================================================================================================================
// The sweep kernel as TWO TEAMS of waves that run their own loops over the nodes of the run and meet only through counters
// in LDS (round 5; the algebra is k_chunk_sweep2's: T_k = G_k F_k, D_L -= F_k^T T_k, F_k+1 = -E^T T_k).
//   D team (waves 0, 1, 2 and 4): the chain  U_k -> G_k = U_k U_k^T -> D~_k+1 = D_k+1 - E^T G_k E -> Cholesky -> U_k+1.
//     Per node: the 15 tiles of G_k into Xg (wave 1 / 2: 48 matrix instructions each, wave 0 / 4: 24 / 20 - these two share a
//     SIMD), the next node built by state pairs straight into Xf (256 threads, <= 3 pairs each), then the three-wave blocked
//     Cholesky (chol80_trio: wave 0 pivot chains, waves 1, 2 panels and trailing products) while wave 4 - the pivot chain's
//     SIMD mate, which therefore carries no matrix work during the chains - streams G_k to HBM.
//   S team (waves 5, 6, 3, 7): the spike of node k as soon as G_k is published.  Wave 5 / 6 hold strip 1 / 2 of the spike
//     (16 columns), wave 3 strips 3 and 0, wave 7 strip 4 (waves 3 and 7 share the SIMD that hosts no D wave: it carries five
//     twelfths of the spike's matrix work).  T strips by spike_gf (F strip in registers, G from Xg), then the tiles of F^T T
//     dealt so that every one is computed where its T strip lives: wave 3 {3,3} {0,3} {0,0}, wave 7 {4,4} {0,4} {3,4},
//     wave 5 {1,1} {0,1} {1,3} {1,4} {1,2}, wave 6 {2,2} {0,2} {2,3} {2,4}; ONE barrier of the four waves, then T -> Y and the
//     stencil pass in place.  Waves 5 and 6 share their SIMDs with the factor helpers and YIELD to them: a helper raises
//     busy[] while it has work and lowers it before every wait, the strip wave polls the word between groups of matrix
//     instructions (the older wave wins the issue arbitration anyway, but every instruction of the younger one that slips
//     into a stall of the helper holds the pipe for 64 cycles).
// Hand-offs (monotonic counters, LDS): gready (G_k in Xg) D -> S; gfdone (a strip wave is through with Xg) S -> D, which then
// overwrites Xg with G_k+1; bvready (next node built: its right-hand side bv[(k+1) & 1] is there) D -> S for the stencil pass;
// fready (a wave's strips of F_k+1 are complete) among the S waves before F^T T reads across strips.  No workgroup barrier
// inside the node loop: the S team's tail (T -> Y, stencil) runs under the D team's G_k+1, the D team's build under the S
// team's T = G F.  LDS-only protocol: the LDS performs one wave's operations in issue order, so a counter update issued behind
// a wave's reads / writes releases them and nothing has to wait for the wave's GLOBAL stores (sub_barrier's fence does).
__device__ __forceinline__ void lds_signal(int* f, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_wait(int* f, int target) {
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_barrier(int* cnt, int& target, int n, int lane) {
  target += n;
  lds_signal(cnt, lane);
  lds_wait(cnt, target);
}
// (k-step 19 - rows 76 .. 79 of F and of T - is skipped everywhere: those rows are exact zeros)
constexpr int SP_STEPS = 4 * NT - 1;
__device__ __forceinline__ void spike_gf3(const double* G, const double (&bF)[NT][4], d4 (&acc)[NT], int li, int lk) {
  double a[3][NT];
  const double* gb = G + lk * LD + li;                 // A(ib, kb)[li][4 s + lk] = G[kb 16 + 4 s + lk][ib 16 + li] (symmetric)
  auto fetch = [&](int buf, int step) {
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) a[buf][ib] = gb[(4 * step) * LD + ib * 16];
  };
#pragma unroll
  for (int ib = 0; ib < NT; ++ib) acc[ib] = d4{0, 0, 0, 0};
  fetch(0, 0);
  fetch(1, 1);
#pragma unroll
  for (int step = 0; step < SP_STEPS; ++step) {
    if (step + 2 < SP_STEPS) fetch((step + 2) % 3, step + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) acc[ib] = mfma(a[step % 3][ib], bF[step >> 2][step & 3], acc[ib]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NQ>
__device__ __forceinline__ void spike_ftt3(d4 (&acc)[NQ], const double* Yf, const d4 (&T)[NT], const int (&as)[NQ], int li, int lk) {
  constexpr int DEPTH = NQ >= 3 ? 2 : (NQ == 2 ? 3 : 6);   // operands requested ~6 matrix instructions ahead of their use
  const double* p = Yf + lk * LD + li;
  double a[DEPTH + 1][NQ];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[d][q] = p[(4 * d) * LD + as[q] * 16];
#pragma unroll
  for (int st = 0; st < SP_STEPS; ++st) {
    if (st + DEPTH < SP_STEPS) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) a[(st + DEPTH) % (DEPTH + 1)][q] = p[(4 * (st + DEPTH)) * LD + as[q] * 16];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = mfma(-a[st % (DEPTH + 1)][q], T[st >> 2][st & 3], acc[q]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// NQ tiles (as[q], j) of F^T T against one T strip, accumulated in the lower tiles of AL (a < j: stored transposed).  The
// accumulators are REQUESTED by al_fetch long before they are needed (ahead of the wave's T = G F: an L2 round trip) and
// folded in by al_run.
template <int NQ>
struct AlTiles {
  d4 acc[NQ];
  int as[NQ], j;
  double* Ag;
  __device__ __forceinline__ double* at(int q, int rr, int li, int lk) const {
    return as[q] >= j ? Ag + (size_t)(as[q] * 16 + 4 * rr + lk) * BS + j * 16 + li
                      : Ag + (size_t)(j * 16 + li) * BS + as[q] * 16 + 4 * rr + lk;
  }
  __device__ __forceinline__ void fetch(int li, int lk) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[q][rr] = *at(q, rr, li, lk);
  }
  __device__ __forceinline__ void run(bool load, const double* Yf, const d4 (&T)[NT], int li, int lk) {
    syrk_mask_n<NQ>(acc, load);
    spike_ftt3<NQ>(acc, Yf, T, as, li, lk);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) *at(q, rr, li, lk) = acc[q][rr];
  }
};
// roles of the two-team sweep by wave (waves w and w + 4 share a SIMD): the pivot-chain wave with a wave that carries no matrix work
// during the chains (0, 4), the two factor helpers together (1, 5), the four strip waves on the other two SIMDs (2, 6 | 3, 7) -
// a wave that streams fp64 matrix instructions holds its SIMD 64 cycles at a time, and every instruction of its SIMD mate,
// vector or matrix, waits for the slot: latency-bound work and matrix streams do not share a SIMD
__device__ __forceinline__ int role8b(int wave) { return (0x75236410u >> (4 * wave)) & 15; }   // {0, 1, 4, 6, 3, 2, 5, 7}
// G = U U^T over the four D waves (0 chain, 1, 2 helpers, 3 the chain's SIMD mate): {0,3,10} {2,4,12,13} {5,8,9,14} {1,6,7,11}:
// 36 | 36 | 32 | 36 matrix instructions, 72 | 68 per SIMD
__device__ __forceinline__ int gram4_tile(int dw, int q) {
  const unsigned v = dw == 0 ? 0xFA30u : (dw == 1 ? 0xDC42u : (dw == 2 ? 0xE985u : 0xB761u));
  const int t = (v >> (4 * q)) & 15;
  return t == 15 ? -1 : t;
}
// G = U U^T over the six builder waves (bw = 0 chain, 1, 2 helpers, 3 the chain's SIMD mate, 4, 5 the strip waves that share
// the helpers' SIMDs): tiles in the nibbles, 15 = none.  {0,10} {1,7} {2,9} {3,6,11} {4,8,12} {5,13,14}: 24 matrix
// instructions each (20 the last), 48 | 48 | 44 per SIMD
__device__ __forceinline__ int gram6_tile(int bw, int q) {
  const unsigned v = bw == 0 ? 0xFA0u : (bw == 1 ? 0xF71u : (bw == 2 ? 0xF92u : (bw == 3 ? 0xB63u : (bw == 4 ? 0xC84u : 0xED5u))));
  const int t = (v >> (4 * q)) & 15;
  return t == 15 ? -1 : t;
}
// The spike of one node on one strip wave (ROLE 4: wave 2, strip 1 | 5: wave 6, strips 2 and 0 | 6: wave 3, strip 3 | 7: wave 7,
// strip 4), a function per role so that each gets its own register allocation.  Tiles of F^T T: role 4 {1,1} {0,1} {1,2},
// role 5 {2,2} {0,2} | {0,0}, role 6 {3,3} {0,3} {1,3} {2,3} {3,4}, role 7 {4,4} {0,4} {1,4} {2,4}.
struct SpikeArgs {
  const double* Xg;
  double* Y;
  double* Ag;
  const double* cRk;
  const double* bvn;
  double* zk;
  int *c_gready, *c_gfdone, *c_fready, *c_bv, *cS;
  int k, n_s;
  bool hasL, has_next;
  long long* stamps;
};
template <int ROLE>
__device__ __forceinline__ void spike_node(const SpikeArgs& A, int& tsb, int ln) {
  constexpr bool two = ROLE == 5;
  constexpr int j0 = ROLE == 4 ? 1 : (ROLE == 5 ? 2 : (ROLE == 6 ? 3 : 4));
  constexpr int NA = ROLE == 4 ? 3 : (ROLE == 5 ? 2 : (ROLE == 6 ? 5 : 4));
  const int li = ln & 15, lk = ln >> 4;
  double* const Y = A.Y;
  auto stamp = [&](int i) { if (A.stamps && ln == 0) A.stamps[i] = (long long)wall_clock64(); };
  AlTiles<NA> al;                                      // tiles against T strip j0
  AlTiles<1> al0;                                      // role 5 only: {0, 0} against T strip 0
  al.Ag = A.Ag;
  al.j = j0;
  if (ROLE == 4) { al.as[0] = 1; al.as[1] = 0; al.as[2 % NA] = 2; }
  if (ROLE == 5) { al.as[0] = 2; al.as[1] = 0; }
  if (ROLE == 6) { al.as[0] = 3; al.as[1] = 0; al.as[2 % NA] = 1; al.as[3 % NA] = 2; al.as[4 % NA] = 4; }
  if (ROLE == 7) { al.as[0] = 4; al.as[1] = 0; al.as[2 % NA] = 1; al.as[3 % NA] = 2; }
  al0.Ag = A.Ag;
  al0.j = 0;
  al0.as[0] = 0;
  if (A.hasL) {
    al.fetch(li, lk);
    if (two) al0.fetch(li, lk);
  }
  lds_wait(A.c_gready, A.k + 1);
  stamp(16 + 4 * (ROLE - 4));
  d4 T0[NT], T1[NT];
  {
    double bF[NT][4];
    const double* yb = Y + lk * LD + j0 * 16 + li;
#pragma unroll
    for (int kb = 0; kb < NT; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) bF[kb][s] = (4 * kb + s < SP_STEPS) ? yb[(kb * 16 + 4 * s) * LD] : 0.0;
    spike_gf3(A.Xg, bF, T0, li, lk);
  }
  if (two) {
    double bF[NT][4];
    const double* yb = Y + lk * LD + li;               // strip 0
#pragma unroll
    for (int kb = 0; kb < NT; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) bF[kb][s] = (4 * kb + s < SP_STEPS) ? yb[(kb * 16 + 4 * s) * LD] : 0.0;
    spike_gf3(A.Xg, bF, T1, li, lk);
  }
  lds_signal(A.c_gfdone, ln);                          // this wave is through with G_k
  stamp(17 + 4 * (ROLE - 4));
  if (A.hasL) {
    lds_wait(A.c_fready, A.n_s * A.k);                 // every strip of F_k is complete
    al.run(A.k > 0, Y, T0, li, lk);
    if (two) al0.run(A.k > 0, Y, T1, li, lk);
  }
  stamp(18 + 4 * (ROLE - 4));
  lds_barrier(A.cS, tsb, A.n_s, ln);                   // nobody reads F_k any more
  // ---- T_k -> Y (own strips), z_k (column 79 of T) -> HBM, then F_k+1 = -E^T T_k in place, column 79 += the next node's
  //      right-hand side
#pragma unroll
  for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, j0, T0[ib], li, lk);
  if (two) {
#pragma unroll
    for (int ib = 0; ib < NT; ++ib) tile_store(Y, ib, 0, T1[ib], li, lk);
  }
  if (ROLE == 7) {
    A.zk[ln] = Y[ln * LD + (BS - 1)];
    if (ln < 16) A.zk[64 + ln] = Y[(64 + ln) * LD + (BS - 1)];
  }
  if (A.has_next) {
    lds_wait(A.c_bv, A.k + 1);                         // the next node's right-hand side is there
    const double* const cRk = A.cRk;
    const double* const bvn = A.bvn;
    const int p = ln % NP, cg = ln / NP;               // cg 0, 1 (lanes 50..63 idle)
#pragma unroll
    for (int h = 0; h < (two ? 2 : 1); ++h) {
      const int c_lo = 16 * (h ? 0 : j0);
      if (cg < 2) {
        const double e00 = cRk[(0 * 3 + 0) * NP + p], e01 = cRk[(0 * 3 + 1) * NP + p], e02 = cRk[(0 * 3 + 2) * NP + p];
        const double e11 = cRk[(1 * 3 + 1) * NP + p], e12 = cRk[(1 * 3 + 2) * NP + p], e22 = cRk[(2 * 3 + 2) * NP + p];
        const double b0 = bvn[p], b1 = bvn[NP + p], b2 = bvn[2 * NP + p];
        double fv[8][3];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int cc = c_lo + cg + 2 * q;
          fv[q][0] = Y[p * LD + cc];
          fv[q][1] = Y[(NP + p) * LD + cc];
          fv[q][2] = Y[(2 * NP + p) * LD + cc];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int cc = c_lo + cg + 2 * q;
          const double f0 = fv[q][0], f1 = fv[q][1], f2 = fv[q][2];
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
  }
  lds_signal(A.c_fready, ln);
  stamp(19 + 4 * (ROLE - 4));
}

constexpr int SW3_VEC = SW_VEC + BS;        // + the second right-hand-side buffer
static constexpr size_t kSweep3Lds = (3 * MAT + SW3_VEC) * sizeof(double);

__global__ void __launch_bounds__(SW_T)
k_chunk_sweep3(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, int* numeric_err, const int* __restrict__ status,
                int m, int n_chunks, int node0, int pin_right) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* const Xf = reinterpret_cast<double*>(smem_raw);   // D~_k -> [L \ U_k] -> D~_k+1
  double* const Xg = Xf + MAT;                               // G_k
  double* const Y = Xg + MAT;                                // spike: F_k -> T_k -> F_k+1; column 79 = right-hand side
  double* const cL = Y + MAT;                          // left tables of the first node; afterwards the odd nodes' right tables
  double* const cR0 = cL + 9 * NP;                     // right tables of the even nodes (of every node when they are uniform)
  double* const bvb = cR0 + 9 * NP;                    // [2][80] right-hand side of the node built last, by node parity
  double* const red = bvb + 2 * BS;                    // [8]
  int* const sync = reinterpret_cast<int*>(red + 8);   // [0] factor trio | [2] helper pair | [12] crit | [13] posts | below
  int* const cB = sync + 4;                            // barrier of the six builder waves
  int* const cS = sync + 5;                            // barrier of the strip waves
  int* const c_gready = sync + 6;
  int* const c_gfdone = sync + 7;
  int* const c_fready = sync + 8;
  int* const c_bv = sync + 9;
  double* const kq = red + 16;                         // [25] each: copies of K.q_w, K.lo, K.hi
  double* const klo = kq + NP;
  double* const khi = klo + NP;
  long long* const lst = reinterpret_cast<long long*>(khi + NP);   // [64] debug stamps, copied out at the end of the kernel
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const FteConst& K = *cst;
  const int c = blockIdx.x;
  const int first = node0 + c * m;
  const bool hasL = c > 0 || node0 > 0, hasR = c + 1 < n_chunks || pin_right;
  const int len = c + 1 < n_chunks ? m : ch.n_nodes - first;       // (the last run: whatever is left, m + 1 at most)
  const int n_int = hasR ? len - 1 : len;
  const int sL = c - 1 + node0, sR = c + node0;                    // separator-chain indices of the run's two ends
  const size_t MB = (size_t)BS * BS;
  const int role = __builtin_amdgcn_readfirstlane(role8b(wave));   // 0 chain, 1 / 2 helpers, 3 chain's mate, 4 .. 7 strip waves
  const bool uni_tables = coupling_tables_uniform(K, first, first + n_int - 1);   // (then the tables of `first` serve every node)
  const int n_s = hasL ? 4 : 1;              // first run: only the right-hand side column (strip 4, role 7) is alive
  const bool builder = role <= 3;            // the D team
  const bool strips = role >= 4 && (hasL || role == 7);
  double gmax_run = 0.0;
  long long* const dbgp = (ch.dbg && (long long)blockIdx.x == ch.dbg[64]) ? ch.dbg : nullptr;
  const int dbg_k = dbgp ? (int)ch.dbg[65] : -1;
  if (dbgp && lane == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    atomicOr(reinterpret_cast<unsigned long long*>(ch.dbg + 67), (unsigned long long)((hw >> 4) & 3) << (4 * wave));
    if (wave == 0) ch.dbg[68] = hw;
  }
#define SW_STAMP(i) do { if (dbgp && k == dbg_k && lane == 0) lst[i] = (long long)wall_clock64(); } while (0)
  if (tid < 16) sync[tid] = 0;
  if (tid < 64) lst[tid] = 0;
  if (c == 0 && node0) {                               // the left pin owns no frames here: its block starts from zero, the
    for (int e = tid; e < BS * BS; e += SW_T) sp.D[e] = 0.0;       // run's spike contribution AL is added by the reduction
    if (tid < BS) sp.b[tid] = 0.0;
  }
  if (tid < NP) {
    kq[tid] = K.q_w[tid];
    klo[tid] = K.lo[tid];
    khi[tid] = K.hi[tid];
  }
  __syncthreads();

  Trio3Sync tsy;
  {  // ---- first node of the run, its spike F_0 = E_l (dense form) with the right-hand side in column 79
    NodeFetch f;
    build_fetch<SW_T>(f, ch, K, first, tid);
    fill_coupling_coef<SW_T>(cL, cR0, K, first, tid, kq);
    for (int e = tid; e < MAT; e += SW_T) Y[e] = 0.0;
    gmax_run = build_finish<SW_T>(Xf, bvb, f, K, first, tid, kq, klo, khi);
    __syncthreads();                                   // node, bv, tables, zeros complete
    if (hasL)
      for (int e = tid; e < 9 * NP; e += SW_T) {
        const int pair = e / NP, p = e % NP, ii = pair / 3, jj = pair % 3;
        if (ii <= jj) Y[(ii * NP + p) * LD + jj * NP + p] = cL[e];
      }
    if (tid < BS) Y[tid * LD + (BS - 1)] = bvb[tid];
    __syncthreads();
    if (role < 3) chol80_trio3(Xf, role, lane, numeric_err, sync, tsy);
    __syncthreads();
  }

  int tb = 0, tsb = 0;                       // rounds of the builders' / the strip waves' barrier
#pragma unroll 1
  for (int k = 0; k < n_int; ++k) {
    const int node = first + k, next = node + 1;
    const bool last = k + 1 == n_int;
    const bool has_next = !last || hasR;
    const int ln = opaque(lane);
    const int li = ln & 15, lk = ln >> 4;
    double* const cRk = (uni_tables || !(k & 1)) ? cR0 : cL;
    if (builder) {
      // ============================== G_k and the next node: the four D waves ==============================
      const int bw = role;                             // 0 chain, 1, 2 helpers, 3 the chain's SIMD mate (wave 4)
      const int bt = bw * 64 + ln;                     // builder thread: state pairs bt, bt + 256, bt + 512 (< 625)
      const int fb_next = 3 * (next - node0);          // (node t holds the local frames 3 (t - pin_left) ..)
      double hq[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, xq[3] = {0, 0, 0}, gq[3] = {0, 0, 0}, cvq[3] = {0, 0, 0}, lamq = 0.0;
      bool liveq[3] = {false, false, false}, ownq[3] = {false, false, false};
      int pdq[3], pcq[3];
      bool ownp[3], dgp[3];
      int pdg = 0;
      bool anydg = false;
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        const int e = bt + 256 * sl;
        ownp[sl] = e < NP * NP;
        pdq[sl] = ownp[sl] ? e / NP : 0;
        pcq[sl] = ownp[sl] ? e % NP : 0;
        dgp[sl] = ownp[sl] && pdq[sl] == pcq[sl];
        if (dgp[sl]) pdg = pdq[sl];                    // (a thread owns at most one diagonal pair: 256 and 512 are no multiples of 26)
        anydg = anydg || dgp[sl];
      }
      if (has_next) {
        const int cur = ch.st->cur;
        const double* Hg = cur ? ch.H1 : ch.H0;
        lamq = ch.st->lam;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (fb_next + j < K.n_frames) {
#pragma unroll
            for (int sl = 0; sl < 3; ++sl)
              if (ownp[sl]) hq[sl][j] = Hg[(size_t)(fb_next + j) * NP * NP + bt + 256 * sl];
          }
        if (anydg) {
          const double* xg = cur ? ch.x1 : ch.x0;
          const double* gg = cur ? ch.g1 : ch.g0;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (fb_next + j < K.n_frames) {
              xq[j] = xg[(size_t)(fb_next + j + HALO) * NP + pdg];
              gq[j] = gg[(size_t)(fb_next + j) * NP + pdg];
            }
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
      if (bw == 0) SW_STAMP(0);
      // U_k complete (the trio came out of the factorisation), wave 4 through with the store of G_k-1; then every strip wave
      // through with Xg
      lds_barrier(cB, tb, 4, ln);
      lds_wait(c_gfdone, n_s * k);
      if (bw == 0) SW_STAMP(1);
      {
        d4 g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = gram4_tile(bw, q);
          if (t >= 0) g[q] = tile_u_ut(Xf, tri_i(t), tri_j(t), li, lk);
        }
        asm volatile("" : "+v"(g[0][0]), "+v"(g[1][0]), "+v"(g[2][0]), "+v"(g[3][0]));
        SW_STAMP(32 + wave);
        if (k > 0 && !uni_tables) fill_coupling_coef<256>(nullptr, cRk, K, node, bt, kq);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = gram4_tile(bw, q);
          if (t >= 0) {
            const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              Xg[(ib * 16 + lk + 4 * rr) * LD + jb * 16 + li] = g[q][rr];
              if (ib != jb) Xg[(jb * 16 + li) * LD + ib * 16 + lk + 4 * rr] = g[q][rr];
            }
          }
        }
      }
      lds_barrier(cB, tb, 4, ln);                      // G_k in Xg, every read of U_k (Xf) done, tables of this node in place
      if (bt == 0) __hip_atomic_fetch_add(c_gready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (bw == 0) SW_STAMP(2);
#pragma unroll
      for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(hq[0][j]), "+v"(hq[1][j]), "+v"(hq[2][j]), "+v"(xq[j]), "+v"(gq[j]));
      asm volatile("" : "+v"(lamq));
      SW_STAMP(24 + wave);
      if (has_next) {
        double* const bvn = bvb + ((k + 1) & 1) * BS;
        // S = E^T G_k E on the thread's blocks (reads Xg), the next node written straight into Xf.  Every LDS read of the three
        // slots first (unconditional: a slot the thread does not own reads pair (0, 0)), then the arithmetic, the stores guarded
        double ca[3][6], cb[3][6], gv[3][3][3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
          const int pr = pdq[sl], pc = pcq[sl];
          ca[sl][0] = cRk[0 * NP + pr]; ca[sl][1] = cRk[1 * NP + pr]; ca[sl][2] = cRk[2 * NP + pr];
          ca[sl][3] = cRk[4 * NP + pr]; ca[sl][4] = cRk[5 * NP + pr]; ca[sl][5] = cRk[8 * NP + pr];
          cb[sl][0] = cRk[0 * NP + pc]; cb[sl][1] = cRk[1 * NP + pc]; cb[sl][2] = cRk[2 * NP + pc];
          cb[sl][3] = cRk[4 * NP + pc]; cb[sl][4] = cRk[5 * NP + pc]; cb[sl][5] = cRk[8 * NP + pc];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            const double* gp = Xg + (jj * NP + pr) * LD + pc;
            gv[sl][jj][0] = gp[0];
            gv[sl][jj][1] = gp[NP];
            gv[sl][jj][2] = gp[2 * NP];
          }
        }
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) asm volatile("" : "+v"(gv[sl][jj][0]), "+v"(gv[sl][jj][1]), "+v"(gv[sl][jj][2]), "+v"(ca[sl][jj]), "+v"(cb[sl][jj + 3]));
        if (bw == 0) SW_STAMP(3);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
          const int pr = pdq[sl], pc = pcq[sl];
          const bool dg = dgp[sl];
          const double a00 = ca[sl][0], a01 = ca[sl][1], a02 = ca[sl][2], a11 = ca[sl][3], a12 = ca[sl][4], a22 = ca[sl][5];
          const double b00 = cb[sl][0], b01 = cb[sl][1], b02 = cb[sl][2], b11 = cb[sl][3], b12 = cb[sl][4], b22 = cb[sl][5];
          double mm[3][3], S[3][3];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            const double g0 = gv[sl][jj][0], g1 = gv[sl][jj][1], g2 = gv[sl][jj][2];
            mm[jj][0] = g0 * b00 + g1 * b01 + g2 * b02;
            mm[jj][1] = g1 * b11 + g2 * b12;
            mm[jj][2] = g2 * b22;
          }
#pragma unroll
          for (int ii = 0; ii < 3; ++ii) {
            S[0][ii] = a00 * mm[0][ii] + a01 * mm[1][ii] + a02 * mm[2][ii];
            S[1][ii] = a11 * mm[1][ii] + a12 * mm[2][ii];
            S[2][ii] = a22 * mm[2][ii];
          }
          double v[3][3];
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) v[j][jp] = (j == jp ? hq[sl][j] : 0.0);
          if (dg) {
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
                if (ownq[j]) gmax_run = fmax(gmax_run, fabs(bb));   // (window sharding: owned frames only)
              }
              v[j][j] = d;
              bvn[j * NP + pr] = bb;
            }
            v[0][1] = v[1][0] = cvq[0];
            v[0][2] = v[2][0] = cvq[1];
            v[1][2] = v[2][1] = cvq[2];
          }
          if (ownp[sl]) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int jp = 0; jp < 3; ++jp) Xf[(j * NP + pr) * LD + jp * NP + pc] = v[j][jp] - S[j][jp];
          }
        }
        // (padding rows / columns 75 .. 79: the factorisation left an identity block there; the diagonal is rewritten all the same,
        //  and the padding of the right-hand side is zero in both buffers)
        if (bt < BS - 3 * NP) {
          Xf[(3 * NP + bt) * LD + 3 * NP + bt] = 1.0;
          bvn[3 * NP + bt] = 0.0;
        }
      }
      SW_STAMP(60 + (wave == 0 ? 0 : (wave == 1 ? 1 : (wave == 5 ? 2 : 3))));
      lds_barrier(cB, tb, 4, ln);                      // next node complete in Xf
      if (bt == 0) __hip_atomic_fetch_add(c_bv, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (bw == 0) SW_STAMP(4);
      if (bw < 3) {
        if (!last) chol80_trio3(Xf, bw, ln, numeric_err, sync, tsy, (dbgp && k == dbg_k) ? lst : nullptr);
        SW_STAMP(8 + wave);
      } else if (bw == 3) {
        // G_k -> HBM, lower tiles only (G is symmetric: the back-substitution mirrors them): the chain's SIMD mate, 30 items per lane
        double* Gg = ch.D + node * MB;
#pragma unroll 6
        for (int q = 0; q < LOWER_ITEMS / 64; ++q) {
          const int idx = ln + 64 * q;
          int r, cc;
          lower_item(idx, r, cc);
          *reinterpret_cast<double2*>(Gg + r * BS + cc) = make_double2(Xg[r * LD + cc], Xg[r * LD + cc + 1]);
        }
        SW_STAMP(8 + wave);
      }
    }
    if (strips) {
      // ============================== the spike of node k ==============================
      SpikeArgs sa;
      sa.Xg = Xg; sa.Y = Y; sa.Ag = sp.AL + (size_t)(hasL ? opaque(sL) : 0) * MB; sa.cRk = cRk; sa.bvn = bvb + ((k + 1) & 1) * BS;
      sa.zk = ch.b + (size_t)node * BS; sa.c_gready = c_gready; sa.c_gfdone = c_gfdone; sa.c_fready = c_fready; sa.c_bv = c_bv; sa.cS = cS;
      sa.k = k; sa.n_s = n_s; sa.hasL = hasL; sa.has_next = has_next; sa.stamps = (dbgp && k == dbg_k) ? lst : nullptr;
      if (role == 4) spike_node<4>(sa, tsb, ln);
      else if (role == 5) spike_node<5>(sa, tsb, ln);
      else if (role == 6) spike_node<6>(sa, tsb, ln);
      else spike_node<7>(sa, tsb, ln);
    }
  }
  __syncthreads();                                     // both teams through
  if (dbgp) {
    if (tid < 64 && lst[tid]) dbgp[tid] = lst[tid];
  }
  if (hasR) {                                          // the node built last is the right separator
    {
      double2* d2 = reinterpret_cast<double2*>(sp.D + (size_t)sR * MB);
      for (int idx = tid; idx < BS * BS / 2; idx += SW_T) {
        const int e = 2 * idx, r = e / BS, cc = e % BS;
        d2[idx] = make_double2(Xf[r * LD + cc], Xf[r * LD + cc + 1]);
      }
    }
    if (tid < BS) sp.b[(size_t)sR * BS + tid] = Y[tid * LD + (BS - 1)];
    if (hasL) {
      double* Cg = sp.Cpl + (size_t)sL * MB;           // block(R, L): rows R, columns L
      for (int e = tid; e < BS * BS; e += SW_T) {
        const int r = e / BS, cc = e % BS;
        Cg[e] = cc < 3 * NP ? Y[r * LD + cc] : 0.0;
      }
    }
  }
#undef SW_STAMP
  publish_gmax<SW_T>(gmax_run, red, ch.gn_part, c, tid);
}

// Separator q: D += AL (the run on its right; lower tiles - the factorisation reads no others).  The right-hand side rode as
// column 79 of the spike, so row 79 of AL holds -(sum W^T y) = the update of b, and rows / columns >= 75 of AL are not
// part of the Schur update.  (A launch of its own: folding it into the sweep - the later of a separator's two runs adds AL -
// needs a device-scope release per workgroup, i.e. an L2 write-back on a multi-XCD part: measured +140 us on the sweep.)
__global__ void __launch_bounds__(256) k_sep_combine(SepView sp, const int* __restrict__ status) {
  if (status && *status != 0) return;
  const int q = blockIdx.x, tid = threadIdx.x;
  const size_t MB = (size_t)BS * BS;
  double* D = sp.D + q * MB;
  const double* A = sp.AL + q * MB;
  // every operand of the thread in flight before the first add (the launch is two HBM round trips, not one per element)
  constexpr int NQ = (LOWER_ITEMS + 255) / 256;
  double2 dv[NQ], av[NQ];
  const double bq = tid < 3 * NP ? sp.b[(size_t)q * BS + tid] : 0.0, aq = tid < 3 * NP ? A[(size_t)(BS - 1) * BS + tid] : 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int idx = tid + 256 * k;
    if (idx < LOWER_ITEMS) {
      int r, cc;
      lower_item(idx, r, cc);
      dv[k] = *reinterpret_cast<const double2*>(D + r * BS + cc);
      av[k] = *reinterpret_cast<const double2*>(A + r * BS + cc);
    }
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int idx = tid + 256 * k;
    if (idx < LOWER_ITEMS) {
      int r, cc;
      lower_item(idx, r, cc);
      if (r < 3 * NP) {                                // rows / columns 75 .. 79 of AL are not part of the Schur update
        const double ax = cc < 3 * NP ? av[k].x : 0.0, ay = cc + 1 < 3 * NP ? av[k].y : 0.0;
        *reinterpret_cast<double2*>(D + r * BS + cc) = make_double2(dv[k].x + ax, dv[k].y + ay);
      }
    }
  }
  if (tid < 3 * NP) sp.b[(size_t)q * BS + tid] = bq + aq;
}

// Back-substitution, one workgroup per run.  x_k = z_k - G_k (E x_k+1) - T_k x_L with T_k = D~_k^-1 F_k; T_k is not
// stored.  Its action on the left separator's solution obeys the recurrence of the spike itself,
//     f_0 = F_0 x_L = E_l x_L,    t_k = T_k x_L = G_k f_k,    f_k+1 = F_k+1 x_L = -E^T t_k,
// so the run is walked twice over the same 31 KB per node (the lower tiles of G_k, mirrored into a symmetric LDS copy):
//   forward  k = 0 .. n-2 :  t_k = G_k f_k, f_k+1 = -E^T t_k            (f_k kept: 80 doubles per node, ch.Wl)
//   backward k = n-1 .. 0 :  x_k = z_k - G_k (E x_k+1 + f_k)
// i.e. <= 61 KB of HBM reads per node instead of the 102 KB of G_k and T_k^T - and no T_k^T store in the sweep (51 KB per node).
// The next node's tiles are requested (registers) before the current node's products and staged into LDS afterwards.
constexpr int BK_T = 512;
constexpr int BK_P = BK_T / BS, BK_W = (BS + BK_P - 1) / BK_P;      // 6 partial sums per row, <= 14 columns each
constexpr int BK_Q = (LOWER_ITEMS + BK_T - 1) / BK_T;      // 4 items per thread (the last round is mostly empty)
__global__ void __launch_bounds__(BK_T)
k_chunk_backsub(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, const int* __restrict__ status, int m,
                int n_chunks, int node0, int pin_right, TrialOut trial) {
  if (status && *status != 0) return;
  __shared__ double Gs[BS * LD], u[BK_P * BK_W], xn[BS], xl[BS], ysc[BK_P * BS], cL[9 * NP], cR[9 * NP];
  const int tid = threadIdx.x;
  if (tid >= BS && tid < BK_P * BK_W) u[tid] = 0.0;      // (the padding behind u stays zero: see product)
  const int c = blockIdx.x, first = node0 + c * m;
  const bool hasL = c > 0 || node0 > 0, hasR = c + 1 < n_chunks || pin_right;
  const int len = c + 1 < n_chunks ? m : ch.n_nodes - first;
  const int n_int = hasR ? len - 1 : len;
  const int sL = c - 1 + node0, sR = c + node0;
  const size_t MB = (size_t)BS * BS;
  const int row = tid % BS, part = tid / BS, c0 = BK_W * part;
  const bool uni_tables = coupling_tables_uniform(*cst, first, first + n_int - 1);   // (one fill serves every node of the run)
  // ---- the trial iterate of the run's frames (fte_api.hip k_trial, same arithmetic): thread r < 75 owns row r of every node
  const int t_cur = ch.st->cur, t_nf = cst->n_frames, t_own_lo = cst->own_lo, t_own_hi = cst->own_hi;
  const double t_lam = ch.st->lam;
  const double* t_x = t_cur ? ch.x1 : ch.x0;
  double* t_xt = const_cast<double*>(t_cur ? ch.x0 : ch.x1);   // (the other iterate buffer: the chain view holds both read-only)
  const double* t_g = t_cur ? ch.g1 : ch.g0;
  const double* t_hd = t_cur ? trial.hd1 : trial.hd0;
  const bool t_thr = tid < 3 * NP;
  const int t_p = tid % NP, t_a = tid / NP;
  const double t_lo = t_thr ? cst->lo[t_p] : 0.0, t_hi = t_thr ? cst->hi[t_p] : 0.0;
  double t_pred = 0.0, t_step = 0.0;
  double t_xv = 0.0, t_gv = 0.0, t_d0 = 0.0;            // operands of the node being solved (requested a node ahead)
  double s_xv = 0.0, s_gv = 0.0, s_d0 = 0.0, s_dx = 0.0;  // ... and of the run's right separator
  auto trial_fetch = [&](int node) {
    const int n = 3 * (node - node0) + t_a;
    if (t_thr && n < t_nf) {
      t_xv = t_x[(size_t)(n + HALO) * NP + t_p];
      t_gv = t_g[(size_t)n * NP + t_p];
      t_d0 = t_hd[(size_t)n * NP + t_p];
    }
  };
  auto trial_row = [&](int node, double delta) {        // (a bound-active variable takes a step of exactly 0: see k_trial)
    const int n = 3 * (node - node0) + t_a;
    if (t_thr && n < t_nf) {
      const double gtol = GRAD_ZERO_REL * t_d0;
      const bool fixed = (t_xv <= t_lo && t_gv > gtol) || (t_xv >= t_hi && t_gv < -gtol);
      const double d = fixed ? 0.0 : delta, pg = fixed ? 0.0 : t_gv;
      const double xnew = fmin(fmax(t_xv + d, t_lo), t_hi);
      t_xt[(size_t)(n + HALO) * NP + t_p] = xnew;
      if (n >= t_own_lo && n < t_own_hi) {              // (window sharding: only owned frames enter the global sums)
        t_pred += 0.5 * d * (t_lam * fmax(t_d0, DIAG_FLOOR) * d - pg);
        t_step = fmax(t_step, fabs(xnew - t_xv));
      }
    }
  };
  double2 gq[BK_Q];
  auto fetch = [&](int node) {                          // lower tiles of G_node -> registers
    const double* G = ch.D + node * MB;
#pragma unroll
    for (int q = 0; q < BK_Q; ++q) {
      const int idx = tid + BK_T * q;
      if (idx < LOWER_ITEMS) {
        int r, cc;
        lower_item(idx, r, cc);
        gq[q] = *reinterpret_cast<const double2*>(G + r * BS + cc);
      }
    }
  };
  auto stage = [&]() {                                  // registers -> symmetric matrix in LDS (strictly-lower tiles mirrored)
#pragma unroll
    for (int q = 0; q < BK_Q; ++q) {
      const int idx = tid + BK_T * q;
      if (idx < LOWER_ITEMS) {
        int r, cc;
        lower_item(idx, r, cc);
        Gs[r * LD + cc] = gq[q].x;
        Gs[r * LD + cc + 1] = gq[q].y;
        if ((cc >> 4) < (r >> 4)) {
          Gs[cc * LD + r] = gq[q].x;
          Gs[(cc + 1) * LD + r] = gq[q].y;
        }
      }
    }
  };
  auto product = [&]() {                                // ysc <- partial sums of Gs u (BK_P per row)
    if (tid < BK_P * BS) {
      // (all BK_W loads unconditional - u is zero behind its 80 entries, the matrix row is clamped -: the last part's four
      //  missing columns as `if (kk < nc)` compiled to four masked blocks with an LDS round trip each)
      double s0 = 0.0;
#pragma unroll
      for (int kk = 0; kk < BK_W; ++kk) s0 += Gs[min(c0 + kk, BS - 1) * LD + row] * u[c0 + kk];
      ysc[tid] = s0;
    }
  };
  auto row_sum = [&](int r) {                           // fixed order
    double v = ysc[r];
#pragma unroll
    for (int q = 1; q < BK_P; ++q) v += ysc[q * BS + r];
    return v;
  };
  if (c == 0 && sp.flags)                               // (k_sep_tail's hand-off flags: clean for the next iteration)
    for (int e = tid; e < sp.n_flags; e += BK_T) sp.flags[e] = 0;
  if (tid < BS) {
    xl[tid] = hasL ? sp.b[(size_t)sL * BS + tid] : 0.0;
    const double xr = hasR ? sp.b[(size_t)sR * BS + tid] : 0.0;
    xn[tid] = xr;
    if (hasR) ch.b[(size_t)(first + n_int) * BS + tid] = xr;      // the separator's solution joins the chain's vector
    if (hasR) {                                         // (the separator's trial row: operands requested here, used at the end)
      trial_fetch(first + n_int);
      s_xv = t_xv;
      s_gv = t_gv;
      s_d0 = t_d0;
      s_dx = xr;
    }
  }
  double* fst = ch.Wl + (size_t)first * BS;             // f_k of this run's nodes
  if (hasL) {
    // ---------------- forward: t_k = G_k f_k, f_k+1 = -E^T t_k ----------------
    fetch(first);
    fill_coupling_coef<BK_T>(cL, cR, *cst, first, tid);
    __syncthreads();                                   // xl, tables
    if (tid < BS) {
      double v = 0.0;
      if (tid < 3 * NP) {
        const int a = tid / NP, p = tid % NP;
        for (int jj = a; jj < 3; ++jj) v += cL[(a * 3 + jj) * NP + p] * xl[jj * NP + p];    // (E_l x_L)[(a, p)]
      }
      u[tid] = v;
      fst[tid] = v;
    }
    stage();
    for (int k = 0; k + 1 < n_int; ++k) {              // (f_n is not needed: the last node takes no forward product)
      const int node = first + k;
      __syncthreads();                                 // Gs, u
      fetch(node + 1);
      if (k > 0 && !uni_tables) fill_coupling_coef<BK_T>(cL, cR, *cst, node, tid);     // (cR of this node: read after the next barrier)
      product();
      __syncthreads();                                 // ysc, cR; every read of Gs and u done
      if (tid < BS) xn[tid] = row_sum(tid);              // t_k (xn is free until the backward pass)
      stage();
      __syncthreads();
      if (tid < BS) {
        double v = 0.0;
        if (tid < 3 * NP) {
          const int a = tid / NP, p = tid % NP;
          for (int bb = a; bb < 3; ++bb) v -= cR[(a * 3 + bb) * NP + p] * xn[bb * NP + p];   // -(E^T t_k)[(a, p)]
        }
        u[tid] = v;
        fst[(size_t)(k + 1) * BS + tid] = v;
      }
    }
    __syncthreads();
    if (tid < BS) xn[tid] = hasR ? sp.b[(size_t)sR * BS + tid] : 0.0;
  }
  // ---------------- backward: x_k = z_k - G_k (E x_k+1 + f_k) ----------------
  // (the last node's tiles are still in LDS after the forward pass; the first run has no forward pass)
  if (!hasL) {
    fetch(first + n_int - 1);
    stage();
  }
  for (int k = n_int - 1; k >= 0; --k) {
    const int node = first + k;
    if (!uni_tables || !hasL) fill_coupling_coef<BK_T>(cL, cR, *cst, node, tid);   // (uniform: filled by the forward pass)
    const double zi = tid < BS ? ch.b[(size_t)node * BS + tid] : 0.0;
    const double fi = (hasL && tid < BS) ? fst[(size_t)k * BS + tid] : 0.0;
    double xtr = 0.0;
    trial_fetch(node);
    if (k > 0) fetch(node - 1);
    __syncthreads();                                   // tables, xn of the previous round, Gs
    if (tid < BS) {
      double v = fi;
      if (tid < 3 * NP) {
        const int a = tid / NP, p = tid % NP;
        for (int ii = 0; ii <= a; ++ii) v += cR[(ii * 3 + a) * NP + p] * xn[ii * NP + p];   // (E_r x_k+1)[(a, p)]
      }
      u[tid] = v;
    }
    __syncthreads();
    product();
    __syncthreads();                                   // ysc; every read of Gs done
    if (tid < BS) {
      const double x = zi - row_sum(tid);
      xn[tid] = x;
      ch.b[(size_t)node * BS + tid] = x;
      xtr = x;
    }
    if (k > 0) stage();
    trial_row(node, xtr);                              // (after the staging: nothing on the node chain waits for it)
  }
  if (hasR) {
    t_xv = s_xv;
    t_gv = s_gv;
    t_d0 = s_d0;
    trial_row(first + n_int, s_dx);
  }
  // the run's share of the predicted reduction (sum) and of the step length (max): waves 0 and 1 hold the 75 rows
  for (int off = 32; off > 0; off >>= 1) {
    t_pred += __shfl_down(t_pred, off, 64);
    t_step = fmax(t_step, __shfl_down(t_step, off, 64));
  }
  __syncthreads();                                     // (ysc is free)
  if ((tid & 63) == 0 && tid < 128) {
    ysc[2 * (tid >> 6)] = t_pred;
    ysc[2 * (tid >> 6) + 1] = t_step;
  }
  __syncthreads();
  if (tid == 0) {
    trial.pred_part[c] = ysc[0] + ysc[2];
    trial.step_part[c] = fmax(ysc[1], ysc[3]);
  }
}

int chunk_set_func_attributes() {
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_sweep),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_sweep2),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_sweep3),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweep3Lds));
  return ACINO_OK;
}

int chunk_reduce(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                 const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                 Profiler* prof) {
  {
    ProfSpan span(prof, PC_CHUNK_SWEEP, s, pl.n_nodes - pl.n_sep);
    static const int variant = [] { const char* e = getenv("ACINO_SWEEP"); return e ? atoi(e) : 3; }();
    if (variant == 3)
      hipLaunchKernelGGL(k_chunk_sweep3, dim3(pl.n_chunks), dim3(SW_T), kSweep3Lds, s, ch, sp, d_c, d_numeric_err, d_status, pl.m,
                         pl.n_chunks, pl.node0, pl.pin_right);
    else if (variant == 1)
      hipLaunchKernelGGL(k_chunk_sweep, dim3(pl.n_chunks), dim3(SW_T), kSweepLds, s, ch, sp, d_c, d_numeric_err, d_status, pl.m,
                         pl.n_chunks, pl.node0, pl.pin_right);
    else
      hipLaunchKernelGGL(k_chunk_sweep2, dim3(pl.n_chunks), dim3(SW_T), kSweepLds, s, ch, sp, d_c, d_numeric_err, d_status, pl.m,
                         pl.n_chunks, pl.node0, pl.pin_right);
  }
  ACINO_LAUNCH_CHECK();
  if (pl.n_sep == 0) return ACINO_OK;
  BcrChain sc = sepch;
  sc.AL0 = nullptr;
  if (bcr_level0_adds_al(sepsch)) {
    sc.AL0 = sp.AL;                  // level 0 of the reduction adds the runs' contributions itself
  } else {
    ProfSpan span(prof, PC_SEP_COMBINE, s, pl.n_sep);
    hipLaunchKernelGGL(k_sep_combine, dim3(pl.n_sep), dim3(256), 0, s, sp, d_status);
  }
  ACINO_LAUNCH_CHECK();
  return bcr_reduce(sc, sepsch, d_c, d_numeric_err, d_status, s, prof);
}

int chunk_backsub(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                  const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                  Profiler* prof, const TrialOut& trial) {
  if (pl.n_sep > 0) {
    int rc = bcr_backsub(sepch, sepsch, d_c, d_status, s, prof, d_numeric_err);
    if (rc) return rc;
  }
  {
    ProfSpan span(prof, PC_CHUNK_BACKSUB, s, pl.n_nodes - pl.n_sep);
    hipLaunchKernelGGL(k_chunk_backsub, dim3(pl.n_chunks), dim3(BK_T), 0, s, ch, sp, d_c, d_status, pl.m, pl.n_chunks, pl.node0,
                       pl.pin_right, trial);
  }
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

}  // namespace acino
