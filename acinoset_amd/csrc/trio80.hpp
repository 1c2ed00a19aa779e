// The blocked Cholesky of an 80 x 80 node by THREE waves that meet only through LDS counters (round 5; used by the chunk
// sweep, csrc/chunk.hip, and by the narrow-level eliminations of the separator chain, csrc/bcr.hip), and the counter
// primitives themselves.
#pragma once
#include "dense80.hpp"

namespace acino {

// ---- LDS counters: the only synchronisation inside the node loop ----------------------------------------------------------------
// The waves of the sweep meet in SUBSETS (the other waves are busy elsewhere and must not be waited for: no s_barrier) through
// monotonic counters in LDS.  The LDS performs one wave's operations in issue order, so a counter update issued behind the wave's
// LDS reads / writes releases them, and LDS accesses issued after a poll has returned are behind it: the protocol orders LDS
// traffic only - which is all the waves share - and never waits for a wave's GLOBAL loads / stores (a workgroup-scope fence
// would: s_waitcnt vmcnt(0)).
__device__ __forceinline__ void lds_signal(int* f, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_wait(int* f, int target) {
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
// two conditions in one poll: both counters are requested together (one LDS round trip per look instead of two loops)
__device__ __forceinline__ void lds_wait2(int* f0, int t0, int* f1, int t1) {
  for (;;) {
    const int a = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int b = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (a >= t0 && b >= t1) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
// barrier among the n waves that share cnt: every participant adds one and waits for the n-th arrival of this round
__device__ __forceinline__ void lds_barrier(int* cnt, int& target, int n, int lane) {
  target += n;
  lds_signal(cnt, lane);
  lds_wait(cnt, target);
}

// ---- the blocked Cholesky of an 80 x 80 node by three waves, split by what the pivot chain needs ---------------------------------
// (dense80.hpp: chol80 describes the factorisation - L in the lower tiles, U = L^-T built alongside in the upper ones.)
// The five 16-pivot chains are a latency chain of ~1.5 us each on ONE wave (role 0), which also multiplies its next panel tile
// and look-ahead tile itself.  Everything else is tile products, and the chain depends on exactly TWO of them per block
// column: (kb+2, kb+1), its next panel tile, and (kb+2, kb+2), its next look-ahead tile.  Until round 4 the three waves met
// in a barrier after every block column, so whatever slowed a helper landed on the chain.  Now the helpers are split by
// deadline and nobody waits for more than he needs:
//   helper L (role 1, the OLDER wave of the SIMD the helpers share: it wins the issue arbitration): the panel tiles below the
//     chain's and the LOWER trailing products - 8 / 7 / 3 / 0 tile products at block column 0 / 1 / 2 / 3 against the chain's ~2 us
//     per column -, every operand of a step requested up front, the chain's two tiles first and signalled (crit);
//   helper U (role 2): the panel and trailing products of the strictly-upper tiles (U = L^-T, needed when the factorisation is
//     over: for G), 8 / 7 / 8 / 7 products, in the gaps the first leaves on their common matrix pipe; it follows the chain's
//     posts and helper L's "panel of column kb stored" and has no deadline before the last block column.  (Block column 0 is
//     the long one for helper L - 3 panel tiles + 9 lower products - and sets the distance it keeps to the chain for the rest
//     of the factorisation: there row 4 of the lower products is helper U's, stored and signalled (culow) before helper L
//     touches row 4 of block column 1.)
// Last block column: the four tiles (t, 4) U_44 wait for U_44 and for helper U's last products, then one barrier of the three.
//   sync[0]: that barrier   [11]: culow   [12]: crit (1 per block column)   [13]: the chain's posts (U_00, then panel kb /
//   U_kb+1,kb+1)   [14]: helper L's panel columns   [15]: helper U through with block column 3
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
  lds_signal(ccrit, lane);
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
struct TrioSync {
  int t3 = 0, posts = 0, crit = 0, lpan = 0, udone = 0, ulow = 0;
};
// (MEET = false: no closing barrier of the three - the caller's next synchronisation point includes them all)
template <bool MEET = true>
__device__ __forceinline__ void chol80_trio(double* Lm, int role, int lane, int* err, int* sync, TrioSync& ts,
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
    lds_signal(cpost, lane);                             // U_00
#pragma unroll 1
    for (int kb = 0; kb < NT - 1; ++kb) {
      if (kb > 0) lds_wait(ccrit, crit0 + kb);       // tiles (kb+1, kb) and (kb+1, kb+1) carry the update of step kb - 1
      panel_tiles<1>(Lm, kb, kb + 1, li, lk);
      lds_signal(cpost, lane);                           // panel tile (kb+1, kb)
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
      lds_signal(cpost, lane);                           // U_kb+1,kb+1
      if (dbg && lane == 0) dbg[41 + 2 * kb] = (long long)wall_clock64();
    }
    lds_wait(cudone, ts.udone + 1);                  // every tile of U carries the update of step 3
    panel_tiles<1>(Lm, NT - 1, 0, li, lk);             // (0, 4) U_44
    __builtin_amdgcn_s_setprio(0);
  } else if (role == 1) {
    // ---------------- helper L ----------------
    __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
    for (int kb = 0; kb < NT - 1; ++kb) {
      lds_wait(cpost, post0 + 2 * kb + 1);           // U_kk
      if (kb == 1) lds_wait(culow, ts.ulow + 1);     // row 4 carries the update of step 0
      if (kb == 0) panel_tiles<3>(Lm, 0, 2, li, lk);
      else if (kb == 1) panel_tiles<2>(Lm, 1, 3, li, lk);
      else if (kb == 2) panel_tiles<1>(Lm, 2, 4, li, lk);
      lds_signal(clpan, lane);                           // panel tiles (kb+2 .., kb) stored
      lds_wait(cpost, post0 + 2 * kb + 2);           // the chain's panel tile (kb+1, kb)
      if (dbg && lane == 0) dbg[56 + kb] = (long long)wall_clock64();
      if (kb == 0) helperL_trailing<0, LowerList<0, 2, 3>>(Lm, li, lk, ccrit, lane);
      else if (kb == 1) helperL_trailing<1>(Lm, li, lk, ccrit, lane);
      else if (kb == 2) helperL_trailing<2>(Lm, li, lk, ccrit, lane);
      else lds_signal(ccrit, lane);
      if (dbg && lane == 0) dbg[48 + kb] = (long long)wall_clock64();
    }
    lds_wait(cpost, post0 + 2 * (NT - 1) + 1);       // U_44
    lds_wait(cudone, ts.udone + 1);
    panel_tiles<2>(Lm, NT - 1, 1, li, lk);             // (1, 4), (2, 4)
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ---------------- helper U ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int kb = 0; kb < NT - 1; ++kb) {
      lds_wait(cpost, post0 + 2 * kb + 1);           // U_kk
      if (kb == 1) panel_tiles<1>(Lm, 1, 0, li, lk);
      else if (kb == 2) panel_tiles<2>(Lm, 2, 0, li, lk);
      else if (kb == 3) panel_tiles<3>(Lm, 3, 0, li, lk);
      lds_wait(cpost, post0 + 2 * kb + 2);           // the chain's panel tile
      lds_wait(clpan, lp0 + kb + 1);                 // helper L's panel tiles
      if (kb == 0) {
        double P[NT][4];
        load_panel<0>(P, Lm, li, lk);
        trail_rest<LowerList<0, 4, 4>, 0, 0>(Lm, P, li, lk);     // (4, 1) .. (4, 4)
        lds_signal(culow, lane);
        trail_rest<UpperList<0>, 0, 0>(Lm, P, li, lk);
      } else if (kb == 1) helperU_trailing<1>(Lm, li, lk);
      else if (kb == 2) helperU_trailing<2>(Lm, li, lk);
      else helperU_trailing<3>(Lm, li, lk);
      if (dbg && lane == 0) dbg[52 + kb] = (long long)wall_clock64();
    }
    lds_signal(cudone, lane);
    lds_wait(cpost, post0 + 2 * (NT - 1) + 1);       // U_44
    panel_tiles<1>(Lm, NT - 1, 3, li, lk);             // (3, 4)
    __builtin_amdgcn_s_setprio(0);
  }
  if (MEET) lds_barrier(c3, ts.t3, 3, lane);
  ts.posts = post0 + 9;
  ts.crit = crit0 + 4;
  ts.lpan = lp0 + 4;
  ts.udone += 1;
  ts.ulow += 1;
}


}  // namespace acino
