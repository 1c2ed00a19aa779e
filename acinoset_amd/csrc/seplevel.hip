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
  const int ns = __builtin_popcount(strips);
  const int cs = ns <= 2 ? 20 * ns : 60 * ((ns + 3) / 4);        // matrix instructions on the busiest wave
  return cs + 20 * ((n_tiles + 3) / 4) + (first ? 40 : 0);        // (workgroup 0 also stores the factor and y)
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
  std::vector<std::vector<int>> bin(T);
  // start: contiguous chunks in an order that keeps tiles sharing strips together (P_l by rows, P_r by rows, X by rows)
  for (size_t k = 0; k < order.size(); ++k) bin[std::min<size_t>(T - 1, k * T / order.size())].push_back(order[k]);
  auto strips = [&](const std::vector<int>& v) {
    unsigned m = 0;
    for (int c : v) m |= slv_strips_of(c);
    return m;
  };
  auto cost = [&](int g) { return slv_cost(strips(bin[g]), (int)bin[g].size(), g == 0); };
  // local search: move one tile out of the most expensive workgroup while that lowers the maximum (then the sum)
  for (int it = 0; it < 400; ++it) {
    int worst = 0;
    for (int g = 1; g < T; ++g)
      if (cost(g) > cost(worst)) worst = g;
    const int cw = cost(worst);
    int best_gain = 0, best_k = -1, best_to = -1;
    for (size_t k = 0; k < bin[worst].size(); ++k) {
      const int code = bin[worst][k];
      std::vector<int> rest = bin[worst];
      rest.erase(rest.begin() + k);
      const int c_rest = slv_cost(strips(rest), (int)rest.size(), worst == 0);
      for (int g = 0; g < T; ++g) {
        if (g == worst) continue;
        std::vector<int> more = bin[g];
        more.push_back(code);
        const int c_more = slv_cost(strips(more), (int)more.size(), g == 0);
        const int gain = cw - std::max(c_rest, c_more);
        if (gain > best_gain) {
          best_gain = gain;
          best_k = (int)k;
          best_to = g;
        }
      }
    }
    if (best_k < 0) break;
    bin[best_to].push_back(bin[worst][best_k]);
    bin[worst].erase(bin[worst].begin() + best_k);
  }
  // who stores a strip of W_l / W_r to HBM (the back-substitution reads them): the first workgroup that computes it
  unsigned stored = 0;
  for (int g = 0; g < T; ++g) {
    int* row = out + (size_t)g * SLV_STRIDE;
    const unsigned m = strips(bin[g]);
    row[0] = (int)m;
    row[1] = (int)(m & ~stored);
    stored |= m;
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
// place in LDS (in place: the wave holds the whole strip of B in bv) and, when asked, to HBM (column 79 - the rider y - as 0)
template <int IB>
__device__ __forceinline__ void slv_row_tile(const double* Lm, const double (&bv)[20], double* Wb, double* __restrict__ Wg, int cc,
                                             int li, int lk) {
  constexpr int NS = 4 * (IB + 1);
  double av[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) av[t] = Lm[(4 * t + lk) * LD + IB * 16 + li];
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < NS; ++t) acc = mfma(av[t], bv[t], acc);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Wb[(IB * 16 + lk + 4 * rr) * LD + cc + li] = acc[rr];
  if (Wg) {
    const bool rider = cc + li == BS - 1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Wg[(IB * 16 + lk + 4 * rr) * BS + cc + li] = rider ? 0.0 : acc[rr];
  }
}
__device__ __forceinline__ void slv_strip_operand(const double* Wb, const double* yv, double (&bv)[20], int cc, int li, int lk) {
  const bool rider = cc + li == BS - 1;           // column 79 carries the node's right-hand side
#pragma unroll
  for (int t = 0; t < 20; ++t) bv[t] = rider ? yv[4 * t + lk] : Wb[(4 * t + lk) * LD + cc + li];
}

__global__ void __launch_bounds__(256)
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
  double2 cl[13], cr[13];
  if (strips & 0x1Fu) fetch_mat(cl, ch.Cpl + (size_t)loc_l * MB, tid);       // block(i, l): rows i, cols l
  if (strips & 0x3E0u) fetch_mat(cr, ch.Cpl + (size_t)loc_r * MB, tid);      // block(r, i): rows r, cols i  (used transposed)
  load_node_sum<256>(Lm, yv, ch, i, fl, tid);
  if (strips & 0x1Fu) stage_mat(WL, cl, tid);
  if (strips & 0x3E0u) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const int idx = tid + 256 * k;
      if (idx < BS * BS / 2) {
        const int e = 2 * idx, rr = e / BS, c = e % BS;
        WR[c * LD + rr] = cr[k].x;
        WR[(c + 1) * LD + rr] = cr[k].y;
      }
    }
  }
  __syncthreads();
  SLV_STAMP(1);
  chol80(Lm, tid, g == 0 ? numeric_err : nullptr);
  SLV_STAMP(2);
  if (g == 0) {                  // y = U^T b -> Y, the factor -> U (what the back-substitution of this node reads)
    if (tid < 3 * BS) {
      const int row = tid % BS, part = tid / BS;
      double yy = 0.0;
      const int c1 = min(27 * part + 27, row + 1);
      for (int c = 27 * part; c < c1; ++c) yy += Lm[c * LD + row] * yv[c];
      ysc[tid] = yy;
    }
    __syncthreads();
    if (tid < BS) ch.Y[(size_t)i * BS + tid] = ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
    store_mat(ch.U + i * MB, Lm, tid);
  }
  SLV_STAMP(3);
  // ---- strips of W = U^T [C_l | C_r], in place --------------------------------------------------------------------------
  const int ns = __popc(strips);
  auto nth_strip = [&](int q) {
    unsigned m = strips;
    for (int k = 0; k < q; ++k) m &= m - 1;
    return __ffs(m) - 1;
  };
  if (ns >= 3) {                 // a whole strip per wave: no hazards between waves, no barriers
    for (int q = wave; q < ns; q += 4) {
      const int s = nth_strip(q), side = s >= 5, cc = 16 * (side ? s - 5 : s);
      double* Wb = side ? WR : WL;
      double* Wg = ((stores >> s) & 1u) ? (side ? ch.Wr : ch.Wl) + i * MB : nullptr;
      double bv[20];
      slv_strip_operand(Wb, yv, bv, cc, li, lk);
      slv_row_tile<4>(Lm, bv, Wb, Wg, cc, li, lk);
      slv_row_tile<3>(Lm, bv, Wb, Wg, cc, li, lk);
      slv_row_tile<2>(Lm, bv, Wb, Wg, cc, li, lk);
      slv_row_tile<1>(Lm, bv, Wb, Wg, cc, li, lk);
      slv_row_tile<0>(Lm, bv, Wb, Wg, cc, li, lk);
    }
  } else {                       // one or two strips: the row tiles of a strip over the waves {4}, {3}, {2, 0}, {1}
    for (int q = 0; q < ns; ++q) {
      const int s = nth_strip(q), side = s >= 5, cc = 16 * (side ? s - 5 : s);
      double* Wb = side ? WR : WL;
      double* Wg = ((stores >> s) & 1u) ? (side ? ch.Wr : ch.Wl) + i * MB : nullptr;
      double bv[20];
      slv_strip_operand(Wb, yv, bv, cc, li, lk);
      __syncthreads();           // every wave holds the strip before any wave overwrites a part of it
      if (wave == 0) slv_row_tile<4>(Lm, bv, Wb, Wg, cc, li, lk);
      else if (wave == 1) slv_row_tile<3>(Lm, bv, Wb, Wg, cc, li, lk);
      else if (wave == 2) {
        slv_row_tile<2>(Lm, bv, Wb, Wg, cc, li, lk);
        slv_row_tile<0>(Lm, bv, Wb, Wg, cc, li, lk);
      } else slv_row_tile<1>(Lm, bv, Wb, Wg, cc, li, lk);
    }
  }
  __syncthreads();
  SLV_STAMP(4);
  // ---- products: this workgroup's tiles, four per wave and batch (the old sums requested before the matrix instructions) ---
  for (int q0 = 0; q0 < nt; q0 += 16) {
    d4 old[4];
    int code[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + wave + 4 * u;
      code[u] = q < nt ? pl[3 + q] : -1;
      old[u] = d4{0, 0, 0, 0};
      if (code[u] >= 0 && code[u] < 50) {
        const int kind = code[u] / 25, ta = (code[u] % 25) / 5, tb = code[u] % 5;
        const int nb = kind == 0 ? l : r;
        if (nb >= 0 && (fl & (kind == 0 ? 4 : 8))) {
          const double* Sg = (kind == 0 ? ch.SR : ch.SL) + (size_t)nb * MB;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) old[u][rr] = Sg[(ta * 16 + lk + 4 * rr) * BS + tb * 16 + li];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (code[u] < 0) continue;
      const int kind = code[u] / 25, ta = (code[u] % 25) / 5, tb = code[u] % 5;
      if ((kind != 1 && l < 0) || (kind != 0 && r < 0)) continue;
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
  hipLaunchKernelGGL(k_sep_level, dim3(8 * a.per), dim3(256), kSepLevelLds, s, ch, a, d_numeric_err, d_status);
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
