// A narrow level of the block cyclic reduction as ONE launch (round 6): elimination AND Schur products.
//
// bcr.hip runs a level as two dependent launches - k_bcr_elim_deep (factor, W = U^T [C_l | C_r] -> HBM) and
// k_bcr_update_deep (reload W of both eliminated neighbours, D_j -= W^T W, new coupling) - ~21 + ~15 us on a chain of <= 128
// eliminated nodes, whatever its length.  The products of an eliminated node i depend on node i alone:
//     P_l = W_l^T W_l (what the left neighbour's diagonal block loses),  P_r = W_r^T W_r,  X = -W_r^T W_l (new block(r, l))
// so the workgroups that hold W_l, W_r in LDS form them right away and the level has no second phase:
//   * nothing here writes the diagonal block of a REMAINING node (two eliminated nodes contribute to it: no ordering, no
//     atomics): the products go to per-side running sums SR[l] (received from the right) and SL[r]; whoever consumes a node
//     later - its own elimination, the isolated level inside k_sep_tail, k_sep_fold for a pin - forms D + AL + SL + SR.  A node
//     has exactly one writer per side and level, so the sums are deterministic;
//   * the right-hand side rides as column 79 of the coupling operands (the padding column): W'[:, 79] = U^T b = y, and row 79 of
//     P = W'^T W' is W^T y - the update of the neighbour's b, for free on the matrix cores (the convention of the chunk sweep's AL);
//   * the new coupling goes to slot n + i of the coupling array (one slot per eliminated node: the siblings of node i still
//     read both old couplings of i while the first of them stores X); the host schedule tracks where every live coupling is;
//   * y goes to its own array Y: b_i is read by every sibling workgroup and must survive the first one that finishes.
// T workgroups per node (T = 256 / nodes, <= 16) repeat the factorisation (the chip is idle; the pivot chain is what costs) and
// split the 55 output tiles; a workgroup computes exactly the strips of W its tiles need (redundantly w.r.t. its siblings, at
// full depth: no partial sums).  The split is a table per T (slv_plan_build), kept in a __device__ array.
// LDS: factor + W_l + W_r = 3 x 80 x 81 doubles (one workgroup per CU).
#include <algorithm>
#include <mutex>

#include "bcr.hpp"
#include "dense80.hpp"
#include "bcr_dev.hpp"
#include "seplevel.hpp"

namespace acino {

// ---- the split of a node's work over its T workgroups ---------------------------------------------------------------
// tile codes: 0 .. 24 P_l(a, b) = 5 a + b (a >= b only), 25 .. 49 P_r, 50 .. 74 X(a, b) (rows: right neighbour, cols: left)
// strips: bit s < 5: columns 16 s .. of W_l; bit 5 + s: of W_r
static int slv_cost(unsigned strips, int n_tiles, bool first) {
  // matrix instructions on the busiest SIMD after the factorisation: a whole strip (60) per wave, the strips dealt to eight
  // waves = four SIMDs; 20 per product tile, tiles dealt the same way
  const int ns = __builtin_popcount(strips);
  return 60 * ((ns + 3) / 4) + 20 * ((n_tiles + 3) / 4) + (first ? 30 : 0);        // (workgroup 0 also stores the factor)
}
static unsigned slv_strips_of(int code) {
  const int kind = code / 25, a = (code % 25) / 5, b = code % 5;
  if (kind == 0) return (1u << a) | (1u << b);
  if (kind == 1) return (32u << a) | (32u << b);
  return (32u << a) | (1u << b);
}
void slv_plan_build(int T, int* out /* [T][SLV_STRIDE] */) {
  std::vector<int> order;
  for (int kind = 0; kind < 2; ++kind)
    for (int a = 0; a < 5; ++a)
      for (int b = 0; b <= a; ++b) order.push_back(25 * kind + 5 * a + b);
  for (int a = 0; a < 5; ++a)
    for (int b = 0; b < 5; ++b) order.push_back(50 + 5 * a + b);
  auto strips = [&](const std::vector<int>& v) {
    unsigned m = 0;
    for (int c : v) m |= slv_strips_of(c);
    return m;
  };
  auto bins_objective = [&](const std::vector<std::vector<int>>& b, long long& mx, long long& sq) {
    mx = 0;
    sq = 0;
    for (int g = 0; g < (int)b.size(); ++g) {
      const long long c = slv_cost(strips(b[g]), (int)b[g].size(), g == 0);
      mx = std::max(mx, c);
      sq += c * c;
    }
  };
  // rows 0 .. 4 into n contiguous groups of nearly equal weight (w: tiles per row); returns the first row of each group + 5
  auto row_groups = [](int n, const int (&w)[5]) {
    std::vector<int> cut(1, 0);
    int tot = 0, acc = 0;
    for (int a = 0; a < 5; ++a) tot += w[a];
    for (int a = 0; a < 5; ++a) {
      acc += w[a];
      while ((int)cut.size() < n && acc * n >= tot * (int)cut.size() && a + 1 < 5 && 5 - (a + 1) >= n - (int)cut.size()) cut.push_back(a + 1);
    }
    while ((int)cut.size() < n) cut.push_back(std::min(4, cut.back() + 1));
    cut.push_back(5);
    return cut;
  };
  // start: contiguous chunks in an order that keeps tiles sharing strips together (P_l by rows, P_r by rows, X by rows) ...
  std::vector<std::vector<int>> bin(T);
  for (size_t k = 0; k < order.size(); ++k) bin[std::min<size_t>(T - 1, k * T / order.size())].push_back(order[k]);
  // ... or the best STRUCTURED split: npl workgroups share P_l by row ranges, npr share P_r, and R x C share X as a grid of row
  // and column ranges (a rectangle of X needs |rows| strips of W_r and |cols| strips of W_l)
  {
    long long best_mx, best_sq;
    bins_objective(bin, best_mx, best_sq);
    const int wp[5] = {1, 2, 3, 4, 5}, wx[5] = {5, 5, 5, 5, 5};
    for (int npl = 1; npl <= 5 && T >= 3; ++npl)
      for (int npr = 1; npr <= 5; ++npr)
        for (int R = 1; R <= 5; ++R)
          for (int C = 1; C <= 5; ++C) {
            if (npl + npr + R * C != T) continue;
            std::vector<std::vector<int>> b(T);
            int g = 0;
            const std::vector<int> cl = row_groups(npl, wp), cr = row_groups(npr, wp), xr = row_groups(R, wx), xc = row_groups(C, wx);
            for (int q = 0; q < npl; ++q, ++g)
              for (int a = cl[q]; a < cl[q + 1]; ++a)
                for (int c = 0; c <= a; ++c) b[g].push_back(5 * a + c);
            for (int q = 0; q < npr; ++q, ++g)
              for (int a = cr[q]; a < cr[q + 1]; ++a)
                for (int c = 0; c <= a; ++c) b[g].push_back(25 + 5 * a + c);
            for (int q = 0; q < R; ++q)
              for (int u = 0; u < C; ++u, ++g)
                for (int a = xr[q]; a < xr[q + 1]; ++a)
                  for (int c = xc[u]; c < xc[u + 1]; ++c) b[g].push_back(50 + 5 * a + c);
            long long mx, sq;
            bins_objective(b, mx, sq);
            if (mx < best_mx || (mx == best_mx && sq < best_sq)) {
              best_mx = mx;
              best_sq = sq;
              bin = b;
            }
          }
    if (T == 2)                       // P_l + the first n rows of X | P_r + the other rows
      for (int n = 0; n <= 5; ++n) {
        std::vector<std::vector<int>> b(2);
        for (int a = 0; a < 5; ++a)
          for (int c = 0; c <= a; ++c) {
            b[0].push_back(5 * a + c);
            b[1].push_back(25 + 5 * a + c);
          }
        for (int a = 0; a < 5; ++a)
          for (int c = 0; c < 5; ++c) b[a < n ? 0 : 1].push_back(50 + 5 * a + c);
        long long mx, sq;
        bins_objective(b, mx, sq);
        if (mx < best_mx || (mx == best_mx && sq < best_sq)) {
          best_mx = mx;
          best_sq = sq;
          bin = b;
        }
      }
  }
  auto cost = [&](int g) { return slv_cost(strips(bin[g]), (int)bin[g].size(), g == 0); };
  // local search on (largest cost, sum of squared costs): the best single move or swap of tiles between two workgroups, until
  // none improves.  (The first term is what the level waits for; the second keeps the search moving across plateaus of it.)
  auto objective = [&](long long& mx, long long& sq) {
    mx = 0;
    sq = 0;
    for (int g = 0; g < T; ++g) {
      const long long c = cost(g);
      mx = std::max(mx, c);
      sq += c * c;
    }
  };
  for (int it = 0; it < 2000; ++it) {
    long long mx0, sq0;
    objective(mx0, sq0);
    long long best_mx = mx0, best_sq = sq0;
    int bg = -1, bk = -1, bh = -1, bj = -1;          // move tile k of g to h (bj < 0) or swap it with tile j of h
    for (int g = 0; g < T; ++g)
      for (size_t k = 0; k < bin[g].size(); ++k)
        for (int h = 0; h < T; ++h) {
          if (h == g) continue;
          for (int j = -1; j < (int)bin[h].size(); ++j) {
            if (j >= 0 && h < g) continue;            // (each swap once)
            const int a = bin[g][k];
            if (j < 0) {
              bin[g].erase(bin[g].begin() + k);
              bin[h].push_back(a);
            } else {
              std::swap(bin[g][k], bin[h][j]);
            }
            long long mx, sq;
            objective(mx, sq);
            if (mx < best_mx || (mx == best_mx && sq < best_sq)) {
              best_mx = mx;
              best_sq = sq;
              bg = g; bk = (int)k; bh = h; bj = j;
            }
            if (j < 0) {
              bin[h].pop_back();
              bin[g].insert(bin[g].begin() + k, a);
            } else {
              std::swap(bin[g][k], bin[h][j]);
            }
          }
        }
    if (bg < 0) break;
    if (bj < 0) {
      const int a = bin[bg][bk];
      bin[bg].erase(bin[bg].begin() + bk);
      bin[bh].push_back(a);
    } else {
      std::swap(bin[bg][bk], bin[bh][bj]);
    }
  }
  // who stores a strip of W_l / W_r to HBM (the back-substitution reads them): of the workgroups that compute it, the one
  // with the fewest strips to store so far
  std::vector<unsigned> own(T, 0u);
  for (int st = 0; st < 10; ++st) {
    int best = -1;
    for (int g = 0; g < T; ++g)
      if ((strips(bin[g]) >> st) & 1u)
        if (best < 0 || __builtin_popcount(own[g]) < __builtin_popcount(own[best])) best = g;
    if (best >= 0) own[best] |= 1u << st;
  }
  for (int g = 0; g < T; ++g) {
    int* row = out + (size_t)g * SLV_STRIDE;
    row[0] = (int)strips(bin[g]);
    row[1] = (int)own[g];
    row[2] = (int)bin[g].size();
    std::sort(bin[g].begin(), bin[g].end());
    for (size_t k = 0; k < bin[g].size(); ++k) row[3 + k] = bin[g][k];
    for (size_t k = bin[g].size(); k + 3 < (size_t)SLV_STRIDE; ++k) row[3 + k] = -1;
  }
}

__device__ int g_slv_plan[(SLV_MAXT + 1) * SLV_MAXT * SLV_STRIDE];    // [T][g][SLV_STRIDE]

// The tables depend on T alone: written once per device (not stream-ordered: called from context creation, never inside a capture).
int slv_upload_plans() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  ACINO_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < 64 && done[dev]) return ACINO_OK;
  std::vector<int> h((size_t)(SLV_MAXT + 1) * SLV_MAXT * SLV_STRIDE, -1);
  for (int T = 1; T <= SLV_MAXT; ++T) slv_plan_build(T, h.data() + (size_t)T * SLV_MAXT * SLV_STRIDE);
  ACINO_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_slv_plan), h.data(), h.size() * sizeof(int)));
  if (dev >= 0 && dev < 64) done[dev] = true;
  return ACINO_OK;
}

// ---- device side ---------------------------------------------------------------------------------------------------------
// (load_node_sum: bcr_dev.hpp)
// one row tile IB of the strip W(:, cc .. cc+15) = U^T B: acc = sum_{k <= IB} U(k, IB)^T B(k, strip); result into the strip's own
// place in LDS (in place: the wave holds the whole strip of B in bv)
template <int IB>
__device__ __forceinline__ void slv_row_tile(const double* Lm, const double (&bv)[20], double* Wb, int cc, int li, int lk) {
  constexpr int NS = 4 * (IB + 1);
  double av[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) av[t] = Lm[(4 * t + lk) * LD + IB * 16 + li];
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < NS; ++t) acc = mfma(av[t], bv[t], acc);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Wb[(IB * 16 + lk + 4 * rr) * LD + cc + li] = acc[rr];
}
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global store in flight (vmcnt(0)) -
// between the rounds of the strip phase that was a round trip to HBM per round
__device__ __forceinline__ void slv_lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// the first 4 NK rows of the strip Wb(:, cc .. cc + 15) as the B operand (what row tiles <= NK - 1 need); column 79 carries the
// node's right-hand side (the rider): a masked fix-up of the four lanes that hold it, only in the strip that has it
template <int NK>
__device__ __forceinline__ void slv_strip_operand(const double* Wb, const double* yv, double (&bv)[20], int cc, int li, int lk) {
#pragma unroll
  for (int t = 0; t < 4 * NK; ++t) bv[t] = Wb[(4 * t + lk) * LD + cc + li];
  if (cc == BS - 16) {
    if (li == 15) {
#pragma unroll
      for (int t = 0; t < 4 * NK; ++t) bv[t] = yv[4 * t + lk];
    }
  }
}
constexpr int SLV_T = 512;       // threads: two waves per SIMD hide each other's LDS latencies in the strip / product phases
constexpr int SLV_V = (BS * BS / 2 + SLV_T - 1) / SLV_T;

__global__ void __launch_bounds__(SLV_T)
k_sep_level(BcrChain ch, SepLevelArgs a, int* __restrict__ numeric_err, const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Lm = reinterpret_cast<double*>(smem_raw);
  double* WL = Lm + MAT;
  double* WR = WL + MAT;
  double* yv = WR + MAT;       // [80] b_i (with every pending update)
  double* ysc = yv + BS;       // [3][80]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  // (as the narrow levels of bcr.hip: workgroup k runs on XCD k % 8; each XCD takes a contiguous range, so the siblings of a
  //  node and the nodes next to it share one L2)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bx = xcd * a.per + slot;
  if (xcd >= a.nx || slot >= a.per || bx >= a.total) return;
  const int ent = bx / a.T, g = bx % a.T;
  const int* en = a.ent + 6 * ent;
  const int i = en[0], l = en[1], r = en[2], fl = en[3], loc_l = en[4], loc_r = en[5];
  const int* pl = g_slv_plan + ((size_t)a.T * SLV_MAXT + g) * SLV_STRIDE;
  unsigned strips = (unsigned)pl[0], stores = (unsigned)pl[1];
  const int nt = pl[2];
  if (l < 0) strips &= ~0x1Fu;
  if (r < 0) strips &= ~0x3E0u;
  const size_t MB = (size_t)BS * BS;
// (debug stamps: workgroup dbg[64] of the level with T = dbg[65] writes wall-clock ticks of its phases into dbg[0..])
#define SLV_STAMP(k) do { if (ch.dbg && tid == 0 && (long long)bx == ch.dbg[64] && (long long)a.T == ch.dbg[65]) ch.dbg[k] = (long long)wall_clock64(); } while (0)
  SLV_STAMP(0);
  // ---- everything this workgroup reads from HBM, requested at once --------------------------------------------------
  double2 cl[SLV_V], cr[SLV_V];
  if (strips & 0x1Fu) fetch_mat<SLV_T>(cl, ch.Cpl + (size_t)loc_l * MB, tid);       // block(i, l): rows i, cols l
  if (strips & 0x3E0u) fetch_mat<SLV_T>(cr, ch.Cpl + (size_t)loc_r * MB, tid);      // block(r, i): rows r, cols i  (used transposed)
  load_node_sum<SLV_T>(Lm, yv, ch, i, fl, tid);
  if (strips & 0x1Fu) stage_mat<SLV_T>(WL, cl, tid);
  if (strips & 0x3E0u) {
#pragma unroll
    for (int k = 0; k < SLV_V; ++k) {
      const int idx = tid + SLV_T * k;
      if (idx < BS * BS / 2) {
        const int e = 2 * idx, rr = e / BS, c = e % BS;
        WR[c * LD + rr] = cr[k].x;
        WR[(c + 1) * LD + rr] = cr[k].y;
      }
    }
  }
  __syncthreads();
  SLV_STAMP(1);
  // ---- factorisation, with the strips of W = U^T [C_l | C_r] riding under it.  Row tile IB of a strip needs column block IB of U
  //      and the strip's rows <= IB: a wave that owns a strip holds the strip's operand in registers from the start, and once
  //      the panel of block IB is done it computes that row tile (in place: no other wave touches those columns) while wave 0
  //      runs the pivots of block IB + 1.  Owners are the six waves of the three SIMDs the pivot chain does not live on; the
  //      chain's SIMD mate (wave 4) takes a seventh strip whole, after the factorisation, when its SIMD is free.  What is left
  //      after the last panel is row tile 4 of every strip.  (Before: all strips after the factorisation, 4.7-6.4 us.)
  const int ns = __popc(strips);
  const int helper = (wave == 0 || wave == 4) ? -1 : (wave < 4 ? wave - 1 : wave - 2);
  const int q0 = (helper >= 0 && helper < ns) ? helper : -1, q1 = (helper >= 0 && helper < 3 && 7 + helper < ns) ? 7 + helper : -1;
  auto strip_at = [&](int q, double*& Wb, int& cc) {
    unsigned m = strips;
    for (int k = 0; k < q; ++k) m &= m - 1;
    const int s = __ffs(m) - 1, side = s >= 5;
    cc = 16 * (side ? s - 5 : s);
    Wb = side ? WR : WL;
  };
  double bv0[20], bv1[20];
  double *Wb0 = WL, *Wb1 = WL;
  int cc0 = 0, cc1 = 0;
  if (q0 >= 0) {
    strip_at(q0, Wb0, cc0);
    slv_strip_operand<5>(Wb0, yv, bv0, cc0, li, lk);
  }
  if (q1 >= 0) {
    strip_at(q1, Wb1, cc1);
    slv_strip_operand<5>(Wb1, yv, bv1, cc1, li, lk);
  }
  auto row_tiles = [&](int kb) {
    if (q0 < 0) return;
    switch (kb) {
      case 0: slv_row_tile<0>(Lm, bv0, Wb0, cc0, li, lk); if (q1 >= 0) slv_row_tile<0>(Lm, bv1, Wb1, cc1, li, lk); break;
      case 1: slv_row_tile<1>(Lm, bv0, Wb0, cc0, li, lk); if (q1 >= 0) slv_row_tile<1>(Lm, bv1, Wb1, cc1, li, lk); break;
      case 2: slv_row_tile<2>(Lm, bv0, Wb0, cc0, li, lk); if (q1 >= 0) slv_row_tile<2>(Lm, bv1, Wb1, cc1, li, lk); break;
      case 3: slv_row_tile<3>(Lm, bv0, Wb0, cc0, li, lk); if (q1 >= 0) slv_row_tile<3>(Lm, bv1, Wb1, cc1, li, lk); break;
      default: slv_row_tile<4>(Lm, bv0, Wb0, cc0, li, lk); if (q1 >= 0) slv_row_tile<4>(Lm, bv1, Wb1, cc1, li, lk); break;
    }
  };
  chol80<SLV_T / 64>(Lm, tid, g == 0 ? numeric_err : nullptr, nullptr, row_tiles);
  SLV_STAMP(2);
  if (wave == 4 && ns > 6) {       // the seventh strip, whole
    double* Wb;
    int cc;
    strip_at(6, Wb, cc);
    double bv[20];
    slv_strip_operand<5>(Wb, yv, bv, cc, li, lk);
    slv_row_tile<4>(Lm, bv, Wb, cc, li, lk);
    slv_row_tile<3>(Lm, bv, Wb, cc, li, lk);
    slv_row_tile<2>(Lm, bv, Wb, cc, li, lk);
    slv_row_tile<1>(Lm, bv, Wb, cc, li, lk);
    slv_row_tile<0>(Lm, bv, Wb, cc, li, lk);
  }
  slv_lds_barrier();
  SLV_STAMP(3);
  if (g == 0 && tid < 256) store_mat(ch.U + i * MB, Lm, tid);      // the factor (what the back-substitution of this node reads)
  // W_l, W_r for the back-substitution, each strip by one of the workgroups that hold it (the rider's column as 0): requested
  // now, they drain under the products
  for (unsigned m = strips & stores; m; m &= m - 1) {
    const int s = __ffs(m) - 1, side = s >= 5, cc = 16 * (side ? s - 5 : s);
    const double* Wb = side ? WR : WL;
    double* Wg = (side ? ch.Wr : ch.Wl) + i * MB;
    double v[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int e = tid + SLV_T * it, k = e >> 4, j = e & 15;
      v[it] = e < BS * 16 ? Wb[k * LD + cc + j] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int e = tid + SLV_T * it, k = e >> 4, j = e & 15;
      if (e < BS * 16) Wg[k * BS + cc + j] = (cc + j == BS - 1) ? 0.0 : v[it];
    }
  }
  // ---- products: this workgroup's tiles, two per wave and batch (the old sums requested before the matrix instructions) ---
  for (int q0 = 0; q0 < nt; q0 += 16) {
    d4 old[2];
    int code[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = q0 + wave + 8 * u;
      int c = q < nt ? pl[3 + q] : -1;
      if (c >= 0) {
        const int kind = c / 25;
        if ((kind != 1 && l < 0) || (kind != 0 && r < 0)) c = -1;             // (an end node has no such neighbour)
      }
      code[u] = c;
      old[u] = d4{0, 0, 0, 0};
      if (c >= 0 && c < 50) {              // running sums: start from what earlier levels left there
        const int kind = c / 25, ta = (c % 25) / 5, tb = c % 5;
        if (fl & (kind == 0 ? 4 : 8)) {
          const double* Sg = (kind == 0 ? ch.SR + (size_t)l * MB : ch.SL + (size_t)r * MB);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) old[u][rr] = Sg[(ta * 16 + lk + 4 * rr) * BS + tb * 16 + li];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (code[u] < 0) continue;
      const int kind = code[u] / 25, ta = (code[u] % 25) / 5, tb = code[u] % 5;
      const double* A = kind == 0 ? WL : WR;          // rows of the result: columns of A
      const double* B = kind == 1 ? WR : WL;
      d4 acc = mma_seq<BS / 4, true>(old[u], A + lk * LD + ta * 16 + li, 4 * LD, B + lk * LD + tb * 16 + li, 4 * LD);
      if (kind == 2) {                                // new block(r, l); the rider's row and column are not part of it
        double* Xg = ch.Cpl + ((size_t)ch.n_nodes + i) * MB;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = ta * 16 + lk + 4 * rr, col = tb * 16 + li;
          Xg[row * BS + col] = (row == BS - 1 || col == BS - 1) ? 0.0 : acc[rr];
        }
      } else {
        double* Sg = (kind == 0 ? ch.SR + (size_t)l * MB : ch.SL + (size_t)r * MB);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Sg[(ta * 16 + lk + 4 * rr) * BS + tb * 16 + li] = acc[rr];
      }
    }
  }
  SLV_STAMP(4);
  if (g == 0) {                  // y = U^T b -> Y
    const bool from_l = (strips >> 4) & 1u, from_r = (strips >> 9) & 1u;
    if (from_l || from_r) {
      if (tid < BS) ch.Y[(size_t)i * BS + tid] = (from_l ? WL : WR)[tid * LD + BS - 1];
    } else {
      if (tid < 3 * BS) {
        const int row = tid % BS, part = tid / BS;
        double yy = 0.0;
        const int c1 = min(27 * part + 27, row + 1);
        for (int c = 27 * part; c < c1; ++c) yy += Lm[c * LD + row] * yv[c];
        ysc[tid] = yy;
      }
      __syncthreads();
      if (tid < BS) ch.Y[(size_t)i * BS + tid] = ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
    }
  }
  SLV_STAMP(5);
#undef SLV_STAMP
}

// D += AL + SL + SR, b += their rows 79 for nodes that outlive the reduction or are handed to kernels that know nothing of the
// sums; the node's right coupling is copied to its home slot.  One workgroup per entry (node, flags, location).
__global__ void __launch_bounds__(256) k_sep_fold(BcrChain ch, const int* __restrict__ ent, const int* __restrict__ status) {
  if (status && *status != 0) return;
  const int tid = threadIdx.x;
  const int i = ent[3 * blockIdx.x], fl = ent[3 * blockIdx.x + 1], loc = ent[3 * blockIdx.x + 2];
  const size_t MB = (size_t)BS * BS;
  double* Dg = ch.D + i * MB;
  const bool hal = ch.AL0 != nullptr, hsl = (fl & 1) != 0, hsr = (fl & 2) != 0;
  const double* Ag = hal ? ch.AL0 + i * MB : Dg;
  const double* Lg = hsl ? ch.SL + i * MB : Dg;
  const double* Rg = hsr ? ch.SR + i * MB : Dg;
  constexpr int NQ = (LOWER_ITEMS + 255) / 256;
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int idx = tid + 256 * k;
    if (idx < LOWER_ITEMS) {
      int rr, cc;
      lower_item(idx, rr, cc);
      if (rr < 3 * NP) {
        const double2 d = *reinterpret_cast<const double2*>(Dg + rr * BS + cc);
        const double2 av = *reinterpret_cast<const double2*>(Ag + rr * BS + cc);
        const double2 lv = *reinterpret_cast<const double2*>(Lg + rr * BS + cc);
        const double2 rv = *reinterpret_cast<const double2*>(Rg + rr * BS + cc);
        const bool in0 = cc < 3 * NP, in1 = cc + 1 < 3 * NP;
        double x = d.x, y = d.y;
        x += (hal && in0) ? av.x : 0.0;
        y += (hal && in1) ? av.y : 0.0;
        x += (hsl && in0) ? lv.x : 0.0;
        y += (hsl && in1) ? lv.y : 0.0;
        x += (hsr && in0) ? rv.x : 0.0;
        y += (hsr && in1) ? rv.y : 0.0;
        *reinterpret_cast<double2*>(Dg + rr * BS + cc) = make_double2(x, y);
      }
    }
  }
  if (tid < 3 * NP) {
    double v = ch.b[(size_t)i * BS + tid];
    v += hal ? Ag[(size_t)(BS - 1) * BS + tid] : 0.0;
    v += hsl ? Lg[(size_t)(BS - 1) * BS + tid] : 0.0;
    v += hsr ? Rg[(size_t)(BS - 1) * BS + tid] : 0.0;
    ch.b[(size_t)i * BS + tid] = v;
  }
  if (loc >= 0 && loc != i) {
    const double2* s2 = reinterpret_cast<const double2*>(ch.Cpl + (size_t)loc * MB);
    double2* d2 = reinterpret_cast<double2*>(ch.Cpl + (size_t)i * MB);
    for (int e = tid; e < BS * BS / 2; e += 256) d2[e] = s2[e];
  }
}

static constexpr size_t kSepLevelLds = (3 * MAT + BS + 3 * BS) * sizeof(double);

int slv_set_func_attributes() {
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sep_level), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kSepLevelLds));
  return slv_upload_plans();
}

int slv_workgroups_per_node(int n_elim) { return std::max(1, std::min(SLV_MAXT, 256 / std::max(n_elim, 1))); }

int slv_launch_level(const BcrChain& ch, const BcrLevel& lv, int* d_numeric_err, const int* d_status, hipStream_t s) {
  SepLevelArgs a;
  a.ent = ch.d_elim6 + 6 * (size_t)lv.e6_off;
  a.T = lv.T;
  a.total = lv.n_elim * lv.T;
  a.nx = std::min(8, (a.total + 31) / 32);
  a.per = (a.total + a.nx - 1) / a.nx;
  hipLaunchKernelGGL(k_sep_level, dim3(8 * a.per), dim3(SLV_T), kSepLevelLds, s, ch, a, d_numeric_err, d_status);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int slv_launch_fold(const BcrChain& ch, const int* d_entries, int n, const int* d_status, hipStream_t s) {
  if (n <= 0) return ACINO_OK;
  hipLaunchKernelGGL(k_sep_fold, dim3(n), dim3(256), 0, s, ch, d_entries, d_status);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

}  // namespace acino

extern "C" int acino_debug_level_split(int T, int32_t* out) {
  using namespace acino;
  ACINO_REQUIRE(T >= 1 && T <= SLV_MAXT && out, "T in 1 .. 16");
  slv_plan_build(T, out);
  return ACINO_OK;
}
