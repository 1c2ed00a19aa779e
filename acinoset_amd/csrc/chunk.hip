// Chunked substructuring of the block-tridiagonal Gauss-Newton system (the step SURVEY.md section 7.6 describes):
// the chain of 80x80 super-blocks is cut into runs of m consecutive nodes; ONE workgroup eliminates the m - 1 interior
// nodes of a run IN ORDER with every operand resident in LDS, the last node of each run is a separator.  In sequential
// order the couplings between consecutive interior nodes stay the constant third-difference stencils E (<= 3 terms per
// entry, never stored); the only dense fill is the "spike" F_k = block(node k, left separator L):
//
//   node k (D~_k, F_k in LDS; the right-hand side b~_k rides as column 79 of F_k):
//        D~_k = L L^T,  U = L^-T             (blocked Cholesky with the inverse factor, three waves: chol80_trio below)
//        G_k = U U^T  (= D~_k^-1)             -> HBM (lower tiles) for the back-substitution
//        T_k = G_k F_k (= D~_k^-1 F_k; column 79: z_k = D~_k^-1 b~_k -> HBM)
//      left separator:   D_L -= F_k^T T_k    (row 79: its right-hand side)
//      next node:        D~_k+1 = D_k+1 - E^T G_k E    F_k+1 = -E^T T_k (+ b_k+1 in column 79)          (E = E_r(k))
//      (after the last interior node the "next node" is the right separator R: F becomes block(R, L))
//   back-substitution, right to left:  x_k = z_k - G_k (E x_k+1) - T_k x_L,  T_k x_L regenerated from G_k (k_chunk_backsub).
//
// (Until round 4 the spike went through W = U^T F, D_L -= W^T W, T = U W: 180 matrix instructions per 16-column strip and two
//  barriers of the strip waves per node; with G_k formed anyway, T = G F and F^T T cost 152 and none.)
// The separators (one per run, n_chunks - 1) are a block-tridiagonal chain with dense couplings: bcr.hip solves it.
#include "chunk.hpp"

#include <algorithm>
#include <cstdlib>

#include "bcr_dev.hpp"
#include "trio80.hpp"

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

// LDS of the sweep kernel: three 80 x 81 matrices (factor workspace | G_k | spike) + tables = 157.9 KB, one workgroup per CU.
//   cL cR | bv x 2 | red | 16 counters | kq klo khi | debug stamps
constexpr int SW_VEC = 18 * NP + 2 * BS + 8 + 8 + 3 * NP + 64;
static constexpr size_t kSweepLds = (3 * MAT + SW_VEC) * sizeof(double);

// The value of x, made opaque to the optimiser: address arithmetic derived from it cannot be hoisted out of the node loop
// (hoisted, the hundreds of per-tile LDS / HBM addresses of this kernel end up spilled to scratch and are reloaded one by
// one in front of the loads that need them).
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

constexpr int SW_T = 512;
// accumulators of a run's FIRST node start from zero (a select behind the unconditional loads: a conditional load compiles to
// a branch per element with a full wait in front)
template <int NQ>
__device__ __forceinline__ void syrk_mask_n(d4 (&acc)[NQ], bool load) {
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[q][rr] = load ? acc[q][rr] : 0.0;
}
// ================================================================================================================
// The sweep kernel: TWO TEAMS of waves that run their own loops over the nodes of a run and meet only through LDS counters.
//   D team (waves 0, 4 | 1, 5): the chain  U_k -> G_k = U_k U_k^T -> D~_k+1 = D_k+1 - E^T G_k E -> Cholesky -> U_k+1.
//     Per node: G_k as a queue of tiles (below), the next node built from the 325 unordered state pairs straight into Xf
//     (256 threads), then chol80_trio - wave 0 the pivot chains, waves 1 and 5 the helpers - while wave 4, the pivot chain's
//     SIMD mate, which therefore carries no matrix work during the chains, streams G_k to HBM.
//   S team (waves 2, 6 | 3, 7): the spike of node k as soon as G_k is complete.  Wave 2 holds strip 1 of the spike (16 columns),
//     wave 6 strips 2 and 0, wave 3 strip 3, wave 7 strip 4.  T strips by spike_gf (the F strip in registers - the accumulator
//     layout of a 16 x 16 tile IS the B-operand layout of the four k-steps over it -, G from Xg), then the tiles of F^T T, every
//     one computed where its T strip lives (spike_node), then T -> Y and the stencil pass in place.
// Who shares a SIMD with whom decides everything here (waves w and w + 4 do): an fp64 matrix instruction holds its SIMD for 64
// cycles, and EVERY instruction of the SIMD mate - vector or matrix - waits for the slot; a wave that streams matrix
// instructions slows a latency-bound mate 2 - 5x (measured: scripts/mfma64_rate.hip, NOTES_perf.md round 5).  So the pivot
// chain shares with a wave that only moves data, the two helpers share with each other (the older one has the deadline), and
// the four strip waves share among themselves.
// Hand-offs (monotonic counters): g_open / g_ticket / g_done: the queue of G_k's tiles; gfdone (a strip wave is through with
// Xg) S -> D, which then overwrites Xg with G_k+1; c_bv (next node built: its right-hand side bv[(k+1) & 1] is there) D -> S for
// the stencil pass; tdone (a wave's tiles of F^T T are finished: the strips it read may be overwritten) and fready (a wave's
// strips of F_k+1 are complete) among the S waves.  No workgroup barrier inside the node loop: the S team's tail (T -> Y,
// stencil) runs under the D team's G_k+1, the D team's build under the S team's T = G F.
// (k-step 19 - rows 76 .. 79 of F and of T - is skipped everywhere: those rows are exact zeros)
constexpr int SP_STEPS = 4 * NT - 1;
__device__ __forceinline__ void spike_gf(const double* G, const double (&bF)[NT][4], d4 (&acc)[NT], int li, int lk) {
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
__device__ __forceinline__ void spike_ftt(d4 (&acc)[NQ], const double* Yf, const d4 (&T)[NT], const int (&as)[NQ], int li, int lk) {
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
    spike_ftt<NQ>(acc, Yf, T, as, li, lk);
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
__device__ __forceinline__ int role8(int wave) { return (0x75236410u >> (4 * wave)) & 15; }   // {0, 1, 4, 6, 3, 2, 5, 7}
// G_k = U_k U_k^T as a QUEUE of its 15 lower tiles, most expensive first (tile t costs 4 (5 - tri_i(t)) matrix instructions):
// whoever is free draws the next ticket - the four D waves as soon as the factorisation is over and the strip waves are
// through with Xg (g_open), the strip waves when they have finished the previous node's spike and would otherwise wait for
// G_k: two more matrix pipes for the tail of the queue.  g_done counts finished tiles: 15 (k + 1) = G_k complete in Xg.
struct GramQueue {
  int *open, *ticket, *done;
};
__device__ __forceinline__ void gram_help(const GramQueue& q, const double* Xf, double* Xg, int k, int li, int lk, int lane) {
  for (;;) {
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(q.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t) - 15 * k;
    if (t >= 15) {
      // (tickets drawn beyond the node's 15 are handed back: the next node's queue starts at 15 (k + 1))
      if (lane == 0) __hip_atomic_fetch_add(q.ticket, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
    const int ib = tri_i(t), jb = tri_j(t);
    const d4 g = tile_u_ut(Xf, ib, jb, li, lk);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      Xg[(ib * 16 + lk + 4 * rr) * LD + jb * 16 + li] = g[rr];
      if (ib != jb) Xg[(jb * 16 + li) * LD + ib * 16 + lk + 4 * rr] = g[rr];
    }
    lds_signal(q.done, lane);
  }
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
  int *c_gfdone, *c_fready, *c_bv, *tdone;           // tdone[0 .. 2]: tile batches finished by roles 4, 6, 7
  GramQueue gq;
  const double* Xf;
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
  // G_k: help with its tiles once the queue is open, then wait for the last one
  lds_wait(A.gq.open, A.k + 1);
  gram_help(A.gq, A.Xf, const_cast<double*>(A.Xg), A.k, li, lk, ln);
  lds_wait(A.gq.done, 15 * (A.k + 1));
  stamp(16 + 4 * (ROLE - 4));
  d4 T0[NT], T1[NT];
  {
    double bF[NT][4];
    const double* yb = Y + lk * LD + j0 * 16 + li;
#pragma unroll
    for (int kb = 0; kb < NT; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) bF[kb][s] = (4 * kb + s < SP_STEPS) ? yb[(kb * 16 + 4 * s) * LD] : 0.0;
    spike_gf(A.Xg, bF, T0, li, lk);
  }
  if (two) {
    double bF[NT][4];
    const double* yb = Y + lk * LD + li;               // strip 0
#pragma unroll
    for (int kb = 0; kb < NT; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) bF[kb][s] = (4 * kb + s < SP_STEPS) ? yb[(kb * 16 + 4 * s) * LD] : 0.0;
    spike_gf(A.Xg, bF, T1, li, lk);
  }
  lds_signal(A.c_gfdone, ln);                          // this wave is through with G_k
  stamp(17 + 4 * (ROLE - 4));
  if (A.hasL) {
    lds_wait(A.c_fready, A.n_s * A.k);                 // every strip of F_k is complete
    al.run(A.k > 0, Y, T0, li, lk);
    if (two) al0.run(A.k > 0, Y, T1, li, lk);
  }
  stamp(18 + 4 * (ROLE - 4));
  // nobody reads this wave's strips of F_k any more: strip 3 is read by its owner only, strip 4 also by role 6 (tile {3,4}),
  // strip 1 by roles 6 and 7, strips 0 and 2 (role 5) by everybody - and role 5 itself reads no strip but its own: nobody waits
  // for the wave with two strips, which finishes its tiles last
  if (A.hasL) {
    if (ROLE != 5) lds_signal(A.tdone + (ROLE == 4 ? 0 : (ROLE == 6 ? 1 : 2)), ln);
    if (ROLE == 7 || ROLE == 4 || ROLE == 5) lds_wait(A.tdone + 1, A.k + 1);
    if (ROLE == 4 || ROLE == 5) lds_wait(A.tdone + 2, A.k + 1);
    if (ROLE == 5) lds_wait(A.tdone + 0, A.k + 1);
  }
  (void)tsb;
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


__global__ void __launch_bounds__(SW_T)
k_chunk_sweep(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, int* numeric_err, const int* __restrict__ status,
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
  int* const sync = reinterpret_cast<int*>(red + 8);   // 16 counters, named below ([0] and [11 .. 15]: the factor trio's)
  int* const cB = sync + 5;                            // barrier of the four D waves
  int* const tdone = sync + 2;                         // [3]: tile batches of F^T T finished by roles 4, 6, 7 (sync[2 .. 4])
  int* const g_done = sync + 6;                        // tiles of G finished (15 per node)
  int* const g_open = sync + 10;                       // queue of node k open: U_k complete, Xg free
  int* const g_ticket = sync + 1;
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
  const int role = __builtin_amdgcn_readfirstlane(role8(wave));   // 0 chain, 1 / 2 helpers, 3 chain's mate, 4 .. 7 strip waves
  const bool uni_tables = coupling_tables_uniform(K, first, first + n_int - 1);   // (then the tables of `first` serve every node)
  const int n_s = hasL ? 4 : 1;              // first run: only the right-hand side column (strip 4, role 7) is alive
  const bool builder = role <= 3;            // the D team
  const bool strips = role >= 4 && (hasL || role == 7);
  double gmax_run = 0.0;
  long long* const dbgp = (ch.dbg && (long long)blockIdx.x == ch.dbg[64] && ch.dbg[66] == 0) ? ch.dbg : nullptr;
  const int dbg_k = dbgp ? (int)ch.dbg[65] : -1;
  if (dbgp && lane == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    atomicOr(reinterpret_cast<unsigned long long*>(ch.dbg + 67), (unsigned long long)((hw >> 4) & 3) << (4 * wave));
    if (wave == 0) ch.dbg[68] = hw;
  }
#define SW_STAMP(i) do { if (dbgp && k == dbg_k && lane == 0) lst[i] = (long long)wall_clock64(); } while (0)
  if (tid < 16) sync[tid] = 0;
  if (tid < 64) lst[tid] = 0;
  if (dbgp && tid == 0) dbgp[70] = (long long)wall_clock64();    // (run-level stamps: 70 start, 60 first node built, 61 factored, 62 loop done, 71 end)
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

  TrioSync tsy;
  {  // ---- first node of the run, its spike F_0 = E_l (dense form) with the right-hand side in column 79
    NodeFetch f;
    build_fetch<SW_T>(f, ch, K, first, tid);
    fill_coupling_coef<SW_T>(cL, cR0, K, first, tid, kq);
    for (int e = tid; e < MAT; e += SW_T) Y[e] = 0.0;
    gmax_run = build_finish<SW_T>(Xf, bvb, f, K, first, tid, kq, klo, khi);
    __syncthreads();                                   // node, bv, tables, zeros complete
    if (dbgp && tid == 0) lst[60] = (long long)wall_clock64();
    if (hasL)
      for (int e = tid; e < 9 * NP; e += SW_T) {
        const int pair = e / NP, p = e % NP, ii = pair / 3, jj = pair % 3;
        if (ii <= jj) Y[(ii * NP + p) * LD + jj * NP + p] = cL[e];
      }
    if (tid < BS) Y[tid * LD + (BS - 1)] = bvb[tid];
    __syncthreads();
    if (role < 3) chol80_trio(Xf, role, lane, numeric_err, sync, tsy);
    __syncthreads();
    if (dbgp && tid == 0) lst[61] = (long long)wall_clock64();
  }

  int tb = 0, tsb = 0;                       // rounds of the builders' barrier
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
      const int bt = bw * 64 + ln;                     // builder thread
      const int fb_next = 3 * (next - node0);          // (node t holds the local frames 3 (t - pin_left) ..)
      // The next node is symmetric: its 3 x 3 frame blocks are built for the 325 UNORDERED state pairs {p, p'} only - pair e (thread
      // bt owns e = bt and e = bt + 256 < 325) is p = e / 13, p' = p + e % 13 mod 25: every cyclic distance 0 .. 12 once - and
      // written to both places (block (p', p) is the transpose).
      double hq[2][3] = {{0, 0, 0}, {0, 0, 0}}, xq[3] = {0, 0, 0}, gq[3] = {0, 0, 0}, cvq[3] = {0, 0, 0}, lamq = 0.0;
      bool liveq[3] = {false, false, false}, ownq[3] = {false, false, false};
      int pdq[2], pcq[2];
      bool ownp[2], dgp[2];
      int pdg = 0;
      bool anydg = false;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int e = bt + 256 * sl;
        ownp[sl] = e < 13 * NP;
        const int pa = ownp[sl] ? e / 13 : 0, dd = ownp[sl] ? e % 13 : 0;
        pdq[sl] = pa;
        pcq[sl] = pa + dd >= NP ? pa + dd - NP : pa + dd;
        dgp[sl] = ownp[sl] && dd == 0;
        if (dgp[sl]) pdg = pa;                         // (a thread owns at most one diagonal pair: e = 13 p, and 256 = 9 mod 13)
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
            for (int sl = 0; sl < 2; ++sl)
              if (ownp[sl]) hq[sl][j] = Hg[(size_t)(fb_next + j) * HPAIRS + bt + 256 * sl];   // (pair e = 13 p + d: as stored)
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
      const GramQueue gqu{g_open, g_ticket, g_done};
      if (bt == 0) __hip_atomic_fetch_add(g_open, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (k > 0 && !uni_tables) fill_coupling_coef<256>(nullptr, cRk, K, node, bt, kq);
      gram_help(gqu, Xf, Xg, k, li, lk, ln);
      SW_STAMP(32 + wave);
      lds_barrier(cB, tb, 4, ln);                      // (the tables of this node in place)
      lds_wait(g_done, 15 * (k + 1));                  // G_k in Xg, every read of U_k (Xf) done
      if (bw == 0) SW_STAMP(2);
#pragma unroll
      for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(hq[0][j]), "+v"(hq[1][j]), "+v"(xq[j]), "+v"(gq[j]));
      asm volatile("" : "+v"(lamq));
      if (has_next) {
        double* const bvn = bvb + ((k + 1) & 1) * BS;
        // S = E^T G_k E on the thread's blocks (reads Xg), the next node written straight into Xf.  Every LDS read of the three
        // slots first (unconditional: a slot the thread does not own reads pair (0, 0)), then the arithmetic, the stores guarded
        double ca[2][6], cb[2][6], gv[2][3][3];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
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
        for (int sl = 0; sl < 2; ++sl) {
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
              for (int jp = 0; jp < 3; ++jp) {
                const double val = v[j][jp] - S[j][jp];
                Xf[(j * NP + pr) * LD + jp * NP + pc] = val;
                if (!dg) Xf[(jp * NP + pc) * LD + j * NP + pr] = val;
              }
          }
        }
        // (padding rows / columns 75 .. 79: the factorisation left an identity block there; the diagonal is rewritten all the same,
        //  and the padding of the right-hand side is zero in both buffers)
        if (bt < BS - 3 * NP) {
          Xf[(3 * NP + bt) * LD + 3 * NP + bt] = 1.0;
          bvn[3 * NP + bt] = 0.0;
        }
      }
      lds_barrier(cB, tb, 4, ln);                      // next node complete in Xf
      if (bt == 0) __hip_atomic_fetch_add(c_bv, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (bw == 0) SW_STAMP(4);
      if (bw < 3) {
        // (no closing barrier of the trio: the four D waves meet at the top of the next iteration, before anyone reads the factor)
        if (!last) chol80_trio<false>(Xf, bw, ln, numeric_err, sync, tsy, (dbgp && k == dbg_k) ? lst : nullptr);
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
      sa.zk = ch.b + (size_t)node * BS; sa.gq = GramQueue{g_open, g_ticket, g_done}; sa.Xf = Xf; sa.c_gfdone = c_gfdone; sa.c_fready = c_fready; sa.c_bv = c_bv; sa.tdone = tdone;
      sa.k = k; sa.n_s = n_s; sa.hasL = hasL; sa.has_next = has_next; sa.stamps = (dbgp && k == dbg_k) ? lst : nullptr;
      if (role == 4) spike_node<4>(sa, tsb, ln);
      else if (role == 5) spike_node<5>(sa, tsb, ln);
      else if (role == 6) spike_node<6>(sa, tsb, ln);
      else spike_node<7>(sa, tsb, ln);
    }
  }
  __syncthreads();                                     // both teams through
  if (dbgp) {
    if (tid == 0) lst[62] = (long long)wall_clock64();
    __syncthreads();
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
  if (dbgp && tid == 0) dbgp[71] = (long long)wall_clock64();
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
// so the run is walked twice over the same 31 KB per node (the lower tiles of G_k):
//   forward  k = 0 .. n-2 :  t_k = G_k f_k, f_k+1 = -E^T t_k            (f_k kept: 80 doubles per node, ch.Wl)
//   backward k = n-1 .. 0 :  x_k = z_k - G_k (E x_k+1 + f_k)
// i.e. 2 n - 1 DEPENDENT 80 x 80 matrix-vector products per run ("steps"), 61 KB of HBM reads per node.  Until round 4 a step
// took 2.1 us - three barriers of eight waves around the product (partial sums through LDS, row sums, stencil by 80 threads) -
// and a prefetch distance of one step.  Now the workgroup is two teams that meet only through LDS counters:
//   * FOUR PRODUCT WAVES, one per SIMD, which touch no global memory but their stores.  Lane (p, part) - p = 8 wave + lane / 8
//     one of the 25 states, part = lane % 8 - multiplies the THREE rows (a, p), a = 0 .. 2 (the state's entry in the node's three
//     frames) by the columns part + 8 j, j = 0 .. 9: 30 multiply-adds, then a three-step butterfly over the 8 lanes of the state
//     (DPP; every lane ends with the three complete sums, bit-identical).  The stencils E couple only the three frames of ONE
//     state, so lanes part = 0 .. 2 turn the sums into the next step's vector entries (a, p) with no further exchange, store
//     x_k / f_k+1 and form the trial iterate of their row; the new vector goes to the other of two LDS buffers, the four waves
//     meet at a counter, next step.
//   * FOUR LOADER WAVES, each with one whole node in flight (its 1920 two-double items = 30 per lane, the row operands z_k and
//     the trial row's x / g / diag H, f_k of far nodes, the step's coupling table where the tables change from node to node):
//     HBM -> registers -> one of two LDS stages when the product waves are through with it -> "ready".  A loader may sit in
//     s_waitcnt vmcnt(0) for as long as HBM takes - nobody waits for IT at a barrier - and the four of them keep 120 KB per CU in
//     flight, which is what the stream needs (the compiler turns a register ring of prefetches inside ONE wave's loop into
//     vmcnt(0 .. 1) waits: measured 66 us against the 51 us of the round-4 kernel).
// The LDS image of a node: lower tiles only, leading dimension 82 (16-byte rows for ds_write_b128; the transposed reads
// G[c][r] of the upper part - stride 164 words - spread over the banks).  f_k of the eight nodes before the turning point come
// from a ring in LDS written by the row lanes themselves (they are read back sooner than a loader could fetch them).
constexpr int BK_PW = 4;                                     // product waves (waves 0 .. 3)
constexpr int BK_NL = 4;                                     // loader waves: step t is loader t % 4's
constexpr int BK_T = 64 * (BK_PW + BK_NL);
constexpr int BK_NB = 2;                                     // LDS stages
constexpr int BLD = 82;
constexpr int BK_Q = LOWER_ITEMS / 64;                       // 30 two-double items per loader lane
constexpr int BK_FL = 8;                                     // f_k of the last BK_FL nodes of the forward pass stay in LDS
static_assert(LOWER_ITEMS == 64 * NT * (NT + 1), "30 rounds of 64 items: 2 (ib + 1) per tile row");
static_assert(BK_FL >= BK_NL + BK_NB + 2, "a far f_k must have been stored (and waited for) before a loader asks for it");
// one stage: the node's tiles | z x g diagH f rows [5][80] | right table of the step's node [225] (+ 1: even size)
constexpr int BK_OPS = BS * BLD, BK_TAB = BK_OPS + 5 * BS, BK_STAGE = BK_TAB + 9 * NP + 1;
static constexpr size_t kBacksubLds =
    (BK_NB * BK_STAGE + 2 * BS + BK_FL * BS + 2 * 9 * NP + 16 + NP + 1) * sizeof(double) + 16 * sizeof(int);

typedef double d2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {      // v + (v of the lane CTRL names)
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// sum over the 8 lanes of a state (i <-> 7 - i, then i <-> i ^ 1, i <-> i ^ 2): every lane receives the same bits (a + b and
// b + a at every level)
__device__ __forceinline__ double lanes8_sum(double v) {
  v = dpp_add<0x141>(v);                                     // row_half_mirror
  v = dpp_add<0xB1>(v);                                      // quad_perm [1, 0, 3, 2]
  return dpp_add<0x4E>(v);                                   // quad_perm [2, 3, 0, 1]
}
__global__ void __launch_bounds__(BK_T)
k_chunk_backsub(BcrChain ch, SepView sp, const FteConst* __restrict__ cst, const int* __restrict__ status, int m,
                int n_chunks, int node0, int pin_right, TrialOut trial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* const stg = reinterpret_cast<double*>(smem_raw);   // [BK_NB][BK_STAGE]
  double* const ub = stg + BK_NB * BK_STAGE;                   // [2][80] the product's vector, by step parity
  double* const fl = ub + 2 * BS;                              // [BK_FL][80] f_k of the nodes before the turning point
  double* const cL0 = fl + BK_FL * BS;                         // left table of the first node
  double* const cRl = cL0 + 9 * NP;                            // right table of the last interior node
  double* const red = cRl + 9 * NP;                            // [16]
  double* const kq = red + 16;                                 // [25] copy of K.q_w (+ 1 pad)
  int* const sync = reinterpret_cast<int*>(kq + NP + 1);       // ready, per loader | stage read | vector written
  int* const c_ready = sync;
  int* const c_rd = sync + BK_NL;
  int* const c_u = sync + BK_NL + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, first = node0 + c * m;
  const bool hasL = c > 0 || node0 > 0, hasR = c + 1 < n_chunks || pin_right;
  const int len = c + 1 < n_chunks ? m : ch.n_nodes - first;
  const int n_int = hasR ? len - 1 : len;
  const int sL = c - 1 + node0, sR = c + node0;
  const size_t MB = (size_t)BS * BS;
  const int nf = hasL && n_int > 0 ? n_int - 1 : 0;         // forward steps; then n_int backward steps
  const int ns = nf + n_int;
  auto node_at = [&](int s) __attribute__((always_inline)) { return s < nf ? s : n_int - 1 - (s - nf); };   // local node of step s
  // ---- loader waves: the tiles of their first node are requested before anything else is looked at (the status word, the
  //      iterate's buffer index, the tables: three dependent round trips to memory that the first tiles now share)
  // Round q of a node's 30 rounds of 64 two-double items: eight rows of one tile row x eight items (one 128-byte line per row),
  // q -> (tile row ib, half h of its rows, 16-column chunk jc <= ib) at compile time.  The lane's part of every address is the
  // same in all rounds (row lane / 8 of the eight, item lane % 8 of the line): two registers, the rest are immediates.
  struct Rounds {
    int row0[BK_Q], col0[BK_Q];
    constexpr Rounds() : row0{}, col0{} {
      int q = 0;
      for (int ib = 0; ib < NT; ++ib)
        for (int h = 0; h < 2; ++h)
          for (int jc = 0; jc <= ib; ++jc) {
            row0[q] = 16 * ib + 8 * h;
            col0[q] = 16 * jc;
            ++q;
          }
    }
  };
  constexpr Rounds RD{};
  const int gl = (lane >> 3) * BS + 2 * (lane & 7), ll = (lane >> 3) * BLD + 2 * (lane & 7);
  d2 g[BK_Q];
  // (rows 75 .. 79 of G_k are rows of the identity - padding - and are never read as rows: the lanes that would fetch them ask
  //  for row 74 again, which costs no traffic: 10 % of the node's bytes)
  const int gl74 = ((lane >> 3) < 3 ? (lane >> 3) : 2) * BS + 2 * (lane & 7);
  auto tile_fetch = [&](int k) __attribute__((always_inline)) {
    const double* G = ch.D + (size_t)(first + k) * MB;
#pragma unroll
    for (int q = 0; q < BK_Q; ++q)
      g[q] = *reinterpret_cast<const d2*>(G + (RD.row0[q] * BS + RD.col0[q]) + (RD.row0[q] + 8 > 3 * NP ? gl74 : gl));
  };
  // Loader lw's jobs: steps lw, lw + BK_NL, ..
  const int lw = wave - BK_PW;
  if (lw >= 0 && lw < ns) tile_fetch(node_at(lw));
  if (status && *status != 0) return;
  const FteConst& K = *cst;
  const bool uni = coupling_tables_uniform(K, first, first + n_int - 1);   // (one table serves every node of the run)
  // the node whose right table step s uses: its own (forward: f_k+1 = -E_k^T t_k) or the next one's (backward: u_k-1 = .. E_k-1 x_k)
  auto table_node = [&](int s) __attribute__((always_inline)) {
    const int k = node_at(s);
    return s < nf ? k : (k > 0 ? k - 1 : 0);
  };
  const int t_cur = ch.st->cur, t_nf = K.n_frames;
  const double* t_x = t_cur ? ch.x1 : ch.x0;
  const double* t_g = t_cur ? ch.g1 : ch.g0;
  const double* t_hd = t_cur ? trial.hd1 : trial.hd0;
  double* const fst = ch.Wl + (size_t)first * BS;           // f_k of this run's nodes
  if (ch.dbg && (long long)blockIdx.x == ch.dbg[64] && ch.dbg[66] == 1 && tid == 0) ch.dbg[61] = (long long)wall_clock64();
  if (c == 0 && sp.flags) {                                 // (k_sep_tail's flags: clean for the next iteration; the int behind
    for (int e = tid; e < sp.n_flags; e += BK_T) sp.flags[e] = 0;   // them is the epoch its hand-off tags are made of)
    if (tid == 0) sp.flags[sp.n_flags] += 1;
  }
  if (tid < 16) sync[tid] = 0;
  if (tid < 2 * BS) ub[tid] = 0.0;
  if (tid < NP) kq[tid] = K.q_w[tid];
  fill_coupling_coef<BK_T>(cL0, nullptr, K, first, tid);
  fill_coupling_coef<BK_T>(nullptr, cRl, K, first + (n_int > 0 ? n_int - 1 : 0), tid);
  double t_pred = 0.0, t_step = 0.0;
  if (wave >= BK_PW) {
    // ================================ loader waves ================================
    // row operands of a backward step: [5][75] values (z, x, g, diag H, f_k-1), entry lane + 64 j; requested a job ahead, with
    // the tiles
    constexpr int NRO = (5 * 3 * NP + 63) / 64;
    double ro[NRO];
    auto ro_fetch = [&](int t) __attribute__((always_inline)) {
      if (t < nf) return;
      const int k = node_at(t), node = first + k;
      const bool far = hasL && k >= 1 && n_int - k >= BK_FL;   // f_k-1 is no longer in the LDS ring
      const int ln = opaque(lane);                          // (the address arithmetic stays inside the loop: registers)
#pragma unroll
      for (int j = 0; j < NRO; ++j) {
        const int e = ln + 64 * j < 5 * 3 * NP ? ln + 64 * j : 5 * 3 * NP - 1, o = e / (3 * NP), r = e % (3 * NP);
        int n = 3 * (node - node0) + r / NP;
        n = n < t_nf ? n : t_nf - 1;
        const size_t nr = (size_t)n * NP + r % NP;
        const double* src = o == 0 ? ch.b + (size_t)node * BS + r
                          : o == 1 ? t_x + nr + (size_t)HALO * NP
                          : o == 2 ? t_g + nr
                          : o == 3 ? t_hd + nr
                                   : fst + (size_t)(far ? k - 1 : 0) * BS + r;
        ro[j] = *src;
      }
      if (lane < BS - 3 * NP) ch.b[(size_t)node * BS + 3 * NP + lane] = 0.0;   // (padding rows of the solution)
    };
    if (lw < ns) ro_fetch(lw);
    __syncthreads();                                        // counters, kq
    for (int t = lw; t < ns; t += BK_NL) {
      const bool bwd = t >= nf;
      long long* const ldbg = (ch.dbg && (long long)blockIdx.x == ch.dbg[64] && ch.dbg[66] == 1 && lane == 0 && t >= 10 && t < 14) ? ch.dbg + 48 + 4 * (t - 10) : nullptr;
      if (ldbg) ldbg[0] = (long long)wall_clock64();
      if (t >= BK_NB) lds_wait(c_rd, BK_PW * (t - BK_NB + 1));   // the product waves are through with the stage
      if (ldbg) { __builtin_amdgcn_s_waitcnt(0x0F70); ldbg[1] = (long long)wall_clock64(); }
      double* const St = stg + (t % BK_NB) * BK_STAGE;
#pragma unroll
      for (int q = 0; q < BK_Q; ++q) *reinterpret_cast<d2*>(St + ll + (RD.row0[q] * BLD + RD.col0[q])) = g[q];
      if (bwd) {
        const int ln = opaque(lane);
#pragma unroll
        for (int j = 0; j < NRO; ++j) {
          const int e = ln + 64 * j;
          if (e < 5 * 3 * NP) St[BK_OPS + (e / (3 * NP)) * BS + e % (3 * NP)] = ro[j];
        }
      }
      // the step's table: the first node's serves every step of a uniform run (the product waves read it in steps 0 and 1);
      // elsewhere most nodes still have the interior table
      if (uni) {
        if (t < BK_NB) right_table_interior<64>(St + BK_TAB, lane, [&](int p) { return kq[p]; });
      } else if (right_table_is_interior(K, first + table_node(t))) {
        right_table_interior<64>(St + BK_TAB, lane, [&](int p) { return kq[p]; });
      } else {
        fill_coupling_coef<64>(nullptr, St + BK_TAB, K, first + table_node(t), lane, kq);
      }
      if (ldbg) { __builtin_amdgcn_s_waitcnt(0xC07F); ldbg[2] = (long long)wall_clock64(); }
      lds_signal(c_ready + lw, lane);
      if (t + BK_NL < ns) {                                 // the next job's tiles and row operands
        tile_fetch(node_at(t + BK_NL));
        ro_fetch(t + BK_NL);
      }
    }
  } else {
    // ================================ product waves ================================
    const int pg = lane >> 3, part = lane & 7;
    const int p = 8 * wave + pg, pc = p < NP ? p : NP - 1;
    const int a = part < 3 ? part : 2, ra = a * NP + pc;    // the lane's row of the triple (lanes part >= 3 shadow row 2)
    const bool rowl = p < NP && part < 3;
    const int t_own_lo = K.own_lo, t_own_hi = K.own_hi;
    const double t_lam = ch.st->lam;
    double* t_xt = const_cast<double*>(t_cur ? ch.x0 : ch.x1);   // (the other iterate buffer: the chain view holds both read-only)
    const double t_lo = K.lo[pc], t_hi = K.hi[pc];
    // trial iterate (fte_api.hip k_trial, same arithmetic): a row lane owns row (a, p) of every node of the run
    auto trial_row = [&](int node, double delta, double xv, double gv, double d0) __attribute__((always_inline)) {
      const int n = 3 * (node - node0) + a;
      if (rowl && n < t_nf) {
        const double gtol = GRAD_ZERO_REL * d0;
        const bool fixed = (xv <= t_lo && gv > gtol) || (xv >= t_hi && gv < -gtol);   // (a bound-active variable takes a step of 0)
        const double d = fixed ? 0.0 : delta, pg_ = fixed ? 0.0 : gv;
        const double xnew = fmin(fmax(xv + d, t_lo), t_hi);
        t_xt[(size_t)(n + HALO) * NP + pc] = xnew;
        if (n >= t_own_lo && n < t_own_hi) {                // (window sharding: only owned frames enter the global sums)
          t_pred += 0.5 * d * (t_lam * fmax(d0, DIAG_FLOOR) * d - pg_);
          t_step = fmax(t_step, fabs(xnew - xv));
        }
      }
    };
    // LDS offsets of the lane's 30 entries: G[(a, p)][part + 8 j] lies in a stored tile when j / 2 <= tile row of (a, p), else it
    // is read as G[part + 8 j][(a, p)]: two bases per row, the rest immediates (30 offsets kept in registers would not fit
    // beside the loaders' 120 registers of tiles: 12 waves share the CU's register file)
    constexpr int NJ = BS / 8;
    int gdir[3], gtrn[3], gtile[3];
#pragma unroll
    for (int aa = 0; aa < 3; ++aa) {
      const int r = aa * NP + pc;
      gdir[aa] = r * BLD + part;
      gtrn[aa] = part * BLD + r;
      gtile[aa] = r >> 4;
    }
    // the two separator solutions, the right separator's trial row, the first vector
    double xr[3] = {0, 0, 0}, xl[3] = {0, 0, 0};
    if (hasR) {
#pragma unroll
      for (int ii = 0; ii < 3; ++ii) xr[ii] = sp.b[(size_t)sR * BS + ii * NP + pc];
      if (tid < BS) ch.b[(size_t)(first + n_int) * BS + tid] = sp.b[(size_t)sR * BS + tid];   // the separator's solution joins the chain's vector
    }
    if (hasL) {
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) xl[jj] = sp.b[(size_t)sL * BS + jj * NP + pc];
    }
    double s_xv, s_gv, s_d0;
    {
      int n = 3 * (first + n_int - node0) + a;
      n = n < t_nf ? n : t_nf - 1;
      s_xv = t_x[(size_t)(n + HALO) * NP + pc];
      s_gv = t_g[(size_t)n * NP + pc];
      s_d0 = t_hd[(size_t)n * NP + pc];
    }
    __syncthreads();                                        // tables, zeros, counters
    double wR = 0.0, f0 = 0.0;                              // (E_r x_R)[(a, p)] for the last interior node, (E_l x_L)[(a, p)]
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
      if (ii <= a) wR += cRl[(ii * 3 + a) * NP + pc] * xr[ii];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
      if (jj >= a) f0 += cL0[(a * 3 + jj) * NP + pc] * xl[jj];
    if (rowl) {
      ub[ra] = nf > 0 ? f0 : f0 + wR;
      if (hasL) {
        fst[ra] = f0;
        fl[ra] = f0;
      }
    }
    lds_signal(c_u, lane);                                  // (the vector of step 0: c_u counts from BK_PW)
    // debug stamps (scripts/backsub_stamps.py): wave 0 of the selected workgroup, per step "stage ready" / "vector ready"
    long long* const bdbg = (ch.dbg && (long long)blockIdx.x == ch.dbg[64] && ch.dbg[66] == 1 && tid == 0) ? ch.dbg : nullptr;
    if (bdbg) bdbg[62] = (long long)wall_clock64();
    double cf[6] = {0, 0, 0, 0, 0, 0};                      // c00 c01 c02 c11 c12 c22 of the step's table for state p
    bool pend = false;                                      // a backward step's stores and trial row, done under the next step's reads
    int p_node = 0;
    double p_xa = 0.0, p_xv = 0.0, p_gv = 0.0, p_d0 = 0.0;
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      const int b = s & 1;
      const bool fwd = s < nf;
      const int k = node_at(s), node = first + k;
      const double* const St = stg + (s % BK_NB) * BK_STAGE;
#define BK_STAMP(i) do { if (bdbg && s >= 8 && s < 16) { __builtin_amdgcn_s_waitcnt(0xC07F); bdbg[6 * (s - 8) + (i)] = (long long)wall_clock64(); } } while (0)
      // the node's stage and the vector (every product wave through with step s - 1), one poll for both
      lds_wait2(c_ready + (s % BK_NL), s / BK_NL + 1, c_u, BK_PW * (s + 1));
      BK_STAMP(0);
      BK_STAMP(1);
      double uv[NJ], gv[3][NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) uv[j] = ub[b * BS + part + 8 * j];
#pragma unroll
      for (int aa = 0; aa < 3; ++aa) {
        const int gd = opaque(gdir[aa]), gt = opaque(gtrn[aa]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) gv[aa][j] = St[(j >> 1) <= gtile[aa] ? gd + 8 * j : gt + 8 * j * BLD];
      }
      if (!uni || s < BK_NB) {
        const double* T = St + BK_TAB;
        cf[0] = T[0 * NP + pc]; cf[1] = T[1 * NP + pc]; cf[2] = T[2 * NP + pc];
        cf[3] = T[4 * NP + pc]; cf[4] = T[5 * NP + pc]; cf[5] = T[8 * NP + pc];
      }
      const double z0 = St[BK_OPS + pc], z1 = St[BK_OPS + NP + pc], z2 = St[BK_OPS + 2 * NP + pc];
      const double o_xv = St[BK_OPS + BS + ra], o_gv = St[BK_OPS + 2 * BS + ra], o_d0 = St[BK_OPS + 3 * BS + ra];
      const double o_f = St[BK_OPS + 4 * BS + ra];
      const double fnear = fl[((k > 0 ? k - 1 : 0) % BK_FL) * BS + ra];
      lds_signal(c_rd, lane);                               // (behind the reads: the LDS serves a wave's requests in order)
      // the previous step's solution row and trial row, while this step's operands are on their way from the LDS
      if (pend) {
        if (rowl) ch.b[(size_t)p_node * BS + ra] = p_xa;
        trial_row(p_node, p_xa, p_xv, p_gv, p_d0);
        pend = false;
      }
      BK_STAMP(2);
      double y[3];
#pragma unroll
      for (int aa = 0; aa < 3; ++aa) {
        double v0 = gv[aa][0] * uv[0], v1 = gv[aa][1] * uv[1];
#pragma unroll
        for (int j = 2; j < NJ; j += 2) {
          v0 = fma(gv[aa][j], uv[j], v0);
          v1 = fma(gv[aa][j + 1], uv[j + 1], v1);
        }
        y[aa] = lanes8_sum(v0 + v1);
      }
      if (bdbg && s >= 8 && s < 16) { asm volatile("" :: "v"(y[0]), "v"(y[1]), "v"(y[2])); bdbg[6 * (s - 8) + 3] = (long long)wall_clock64(); }
      if (fwd) {
        // t_k = y;  f_k+1 = -E^T t_k:  row (a, p) = -sum_{bb >= a} c[a][bb] t[(bb, p)]
        const double f0_ = -(cf[0] * y[0] + cf[1] * y[1] + cf[2] * y[2]);
        const double f1_ = -(cf[3] * y[1] + cf[4] * y[2]);
        const double f2_ = -(cf[5] * y[2]);
        const double fn = a == 0 ? f0_ : (a == 1 ? f1_ : f2_);
        if (rowl) {
          ub[(b ^ 1) * BS + ra] = s + 1 == nf ? fn + wR : fn;
          fl[((k + 1) % BK_FL) * BS + ra] = fn;
          fst[(size_t)(k + 1) * BS + ra] = fn;
        }
        // every f_k of the run stored (at L2) before a loader may ask for one: the loaders start on far f_k only after the product
        // waves have read two stages beyond the turning point
        if (s + 1 == nf) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        lds_signal(c_u, lane);
        BK_STAMP(4);
      } else {
        const double x0 = z0 - y[0], x1 = z1 - y[1], x2 = z2 - y[2];
        const double xa = a == 0 ? x0 : (a == 1 ? x1 : x2);
        // u_k-1 = f_k-1 + E_r(k-1) x_k:  row (a, p) = sum_{ii <= a} c[ii][a] x[(ii, p)]
        const double e0 = cf[0] * x0, e1 = cf[1] * x0 + cf[3] * x1, e2 = cf[2] * x0 + cf[4] * x1 + cf[5] * x2;
        const double fk = hasL ? (n_int - k < BK_FL ? fnear : o_f) : 0.0;
        if (rowl && k > 0) ub[(b ^ 1) * BS + ra] = fk + (a == 0 ? e0 : (a == 1 ? e1 : e2));
        lds_signal(c_u, lane);                              // (the next step's vector is complete: the rest is stores)
        BK_STAMP(4);
        pend = true;
        p_node = node;
        p_xa = xa;
        p_xv = o_xv;
        p_gv = o_gv;
        p_d0 = o_d0;
      }
      if (bdbg && s >= 8 && s < 16) bdbg[6 * (s - 8) + 5] = (long long)wall_clock64();
    }
#undef BK_STAMP
    if (pend) {
      if (rowl) ch.b[(size_t)p_node * BS + ra] = p_xa;
      trial_row(p_node, p_xa, p_xv, p_gv, p_d0);
    }
    if (bdbg) bdbg[63] = (long long)wall_clock64();
    if (hasR) trial_row(first + n_int, a == 0 ? xr[0] : (a == 1 ? xr[1] : xr[2]), s_xv, s_gv, s_d0);
  }
  // the run's share of the predicted reduction (sum) and of the step length (max), fixed order
  for (int off = 32; off > 0; off >>= 1) {
    t_pred += __shfl_down(t_pred, off, 64);
    t_step = fmax(t_step, __shfl_down(t_step, off, 64));
  }
  if (lane == 0 && wave < BK_PW) {
    red[2 * wave] = t_pred;
    red[2 * wave + 1] = t_step;
  }
  __syncthreads();
  if (tid == 0) {
    double sp_ = 0.0, sm = 0.0;
#pragma unroll
    for (int w = 0; w < BK_PW; ++w) {
      sp_ += red[2 * w];
      sm = fmax(sm, red[2 * w + 1]);
    }
    trial.pred_part[c] = sp_;
    trial.step_part[c] = sm;
  }
}

int chunk_set_func_attributes() {
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_sweep),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chunk_backsub),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubLds));
  return ACINO_OK;
}

int chunk_reduce(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                 const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                 Profiler* prof) {
  {
    ProfSpan span(prof, PC_CHUNK_SWEEP, s, pl.n_nodes - pl.n_sep);
    hipLaunchKernelGGL(k_chunk_sweep, dim3(pl.n_chunks), dim3(SW_T), kSweepLds, s, ch, sp, d_c, d_numeric_err, d_status, pl.m,
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
    hipLaunchKernelGGL(k_chunk_backsub, dim3(pl.n_chunks), dim3(BK_T), kBacksubLds, s, ch, sp, d_c, d_status, pl.m, pl.n_chunks, pl.node0,
                       pl.pin_right, trial);
  }
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

}  // namespace acino
