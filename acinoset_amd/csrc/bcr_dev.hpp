// Device helpers shared by the block cyclic reduction (bcr.hip) and the chunk sweep (chunk.hip): the damped Gauss-Newton
// node built in LDS straight from the assembly output, the per-node gradient norm, the third-difference coupling tables.
#pragma once
#include "bcr.hpp"
#include "dense80.hpp"

namespace acino {

// lower-triangular tile enumeration t -> (ib, jb), t = ib (ib + 1) / 2 + jb  (arithmetic, not a table: see dense80.hpp)
__device__ __forceinline__ int tri_i(int t) { return (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10); }
__device__ __forceinline__ int tri_j(int t) {
  const int i = tri_i(t);
  return t - (i * (i + 1)) / 2;
}

// The 15 lower 16x16 tiles of an 80 x 80 row-major matrix as 1920 two-double items: item idx -> (row, even column).
// Tile row ib holds 16 rows of 8 (ib + 1) items each; cumulative item counts 0, 128, 384, 768, 1280, 1920.
constexpr int LOWER_ITEMS = 1920;
__device__ __forceinline__ void lower_item(int idx, int& row, int& col) {
  const int ib = (idx >= 128) + (idx >= 384) + (idx >= 768) + (idx >= 1280);
  const int rem = idx - 64 * ib * (ib + 1), w = 8 * (ib + 1);      // 64 ib (ib + 1) = items in the tile rows above
  row = 16 * ib + rem / w;
  col = 2 * (rem % w);
}

// Damped Gauss-Newton block of chain node t, built in LDS (leading dimension LD) straight from the
// assembly's H/g (no set-up pass through HBM): D = H_gn + lam*diag(H_gn), bound-active variables
// pinned by a 2^70 diagonal boost, identity on padding / non-existent frames; bv = -g (0 where pinned).
// Returns this thread's max |projected gradient| contribution.  All 256 threads; one barrier inside (the
// caller's publish_gmax barrier completes the block).  (Two barriers inside.)
// The global reads of one node, issued up front (a caller may issue them long before the node is built).
struct NodeFetch {
  double hv[8];
  double xv, gv, lam;
  bool row_live;
};
// (NTH = threads of the workgroup: 256, or 512 in the eight-wave chunk sweep)
template <int NTH = 256>
static __device__ __forceinline__ void build_fetch(NodeFetch& f, const BcrChain& ch, const FteConst& K, int t, int tid) {
  const int cur = ch.st->cur;
  const double* x = cur ? ch.x1 : ch.x0;
  const double* g = cur ? ch.g1 : ch.g0;
  const double* H = cur ? ch.H1 : ch.H0;
  f.lam = ch.st->lam;
  const bool sep_left = K.pin_left && t == 0;
  const int fbase = 3 * (t - K.pin_left);
  // the three 25x25 Gauss-Newton blocks (<= 8 entries per thread) and, for the 75 row-owner threads, the state and
  // gradient entry of their row
#pragma unroll
  for (int k = 0; k < 2048 / NTH; ++k) {
    const int idx = tid + NTH * k;
    f.hv[k] = 0.0;
    if (!sep_left && idx < 3 * NP * NP) {
      const int n = fbase + idx / (NP * NP);
      if (n < K.n_frames) f.hv[k] = H[(size_t)n * HPAIRS + hpair((idx % (NP * NP)) / NP, idx % NP)];   // (stored as unordered pairs)
    }
  }
  f.xv = 0.0;
  f.gv = 0.0;
  f.row_live = false;
  if (!sep_left && tid < 3 * NP) {
    const int n = fbase + tid / NP, p = tid % NP;
    if (n < K.n_frames) {
      f.row_live = true;
      f.xv = x[(size_t)(n + HALO) * NP + p];
      f.gv = g[(size_t)n * NP + p];
    }
  }
}
// qw / lo / hi: the per-state weight and bound tables (K.q_w, K.lo, K.hi, or a copy of them in LDS)
template <int NTH = 256>
static __device__ __forceinline__ double build_finish(double* Dm, double* bv, const NodeFetch& f, const FteConst& K, int t,
                                                      int tid, const double* qw, const double* lo, const double* hi,
                                                      long long* dbg = nullptr) {
  const bool sep_left = K.pin_left && t == 0;
  const int fbase = 3 * (t - K.pin_left);
  // (2) structure that needs no memory: zeros, identity padding, intra-node third-difference couplings
  for (int e = tid; e < BS * LD; e += NTH) Dm[e] = 0.0;
  if (dbg && tid == 0) dbg[24] = (long long)wall_clock64();
  __syncthreads();
  if (dbg && tid == 0) dbg[25] = (long long)wall_clock64();
  if (!sep_left) {
    if (tid < BS) {
      const int nfr = fbase + tid / NP;
      if (tid >= 3 * NP || nfr >= K.n_frames) Dm[tid * LD + tid] = 1.0;
    } else if (tid < BS + 3 * NP) {
      const int q = tid - BS, pr = q / NP, p = q % NP;          // frame pairs (0,1), (0,2), (1,2)
      const int ii = pr == 2 ? 1 : 0, jj = pr == 0 ? 1 : 2;
      if (fbase + jj < K.n_frames) {
        const double v = 2.0 * qw[p] * band_coef_clip(K.n_offset + fbase + ii, jj - ii, K.n_global, K.clip_len);
        Dm[(ii * NP + p) * LD + jj * NP + p] = v;
        Dm[(jj * NP + p) * LD + ii * NP + p] = v;
      }
    }
  }
  if (dbg && tid == 0) dbg[26] = (long long)wall_clock64();
  // (3) drop the H blocks in (their targets are disjoint from the entries written in (2))
#pragma unroll
  for (int k = 0; k < 2048 / NTH; ++k) {
    const int idx = tid + NTH * k;
    if (!sep_left && idx < 3 * NP * NP) {
      const int ii = idx / (NP * NP), rem = idx % (NP * NP), p = rem / NP, pc = rem % NP;
      if (fbase + ii < K.n_frames) Dm[(ii * NP + p) * LD + ii * NP + pc] = f.hv[k];
    }
  }
  if (dbg && tid == 0) dbg[28] = (long long)wall_clock64();
  __syncthreads();
  if (dbg && tid == 0) dbg[31] = (long long)wall_clock64();
  // (4) Marquardt damping and pinning of the diagonal, right-hand side, projected-gradient norm
  double gmax = 0.0;
  if (tid < BS) {
    double b = 0.0;
    if (f.row_live) {
      const int p = tid % NP;
      double d = Dm[tid * LD + tid];
      const double gtol = GRAD_ZERO_REL * d;
      const bool fixed = (f.xv <= lo[p] && f.gv > gtol) || (f.xv >= hi[p] && f.gv < -gtol);
      d = d + f.lam * fmax(d, DIAG_FLOOR);
      if (fixed) d *= FIX_SCALE;
      Dm[tid * LD + tid] = d;
      b = fixed ? 0.0 : -f.gv;
      const int nloc = fbase + tid / NP;
      gmax = (nloc >= K.own_lo && nloc < K.own_hi) ? fabs(b) : 0.0;   // (window sharding: owned frames only)
    }
    bv[tid] = b;
  }
  return gmax;
}
static __device__ double build_node(double* Dm, double* bv, const BcrChain& ch, const FteConst& K, int t, int tid) {
  NodeFetch f;
  build_fetch(f, ch, K, t, tid);
  return build_finish(Dm, bv, f, K, t, tid, K.q_w, K.lo, K.hi);
}

// max over the workgroup -> gn_part[node]
template <int NTH = 256>
static __device__ void publish_gmax(double gmax, double* red, double* gn_part, int node, int tid) {
  for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_down(gmax, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = gmax;
  __syncthreads();
  if (tid == 0) {
    double v = red[0];
#pragma unroll
    for (int w = 1; w < NTH / 64; ++w) v = fmax(v, red[w]);
    gn_part[node] = v;
  }
}

// Level-0 couplings are the constant third-difference blocks E (<= 3 non-zeros per column, all on the
// state's own diagonal), so W = U^T E needs no GEMM: column (jj,p) of W is a combination of <= 3 ROWS of U.
//   left  (neighbour i-1): W_l[r][(jj,p)] = sum_{ii<=jj} U[(ii,p)][r] * 2 q_p band(f_i-3+jj, 3+ii-jj)
//   right (neighbour i+1): W_r[r][(ii,p)] = sum_{jj>=ii} U[(jj,p)][r] * 2 q_p band(f_i+jj,   3+ii-jj)
// coef[(ii*3 + jj) * NP + p], ii <= jj: the two constant coupling blocks of node i (left: columns in node i-1,
// right: columns in node i+1).  450 doubles, filled once per workgroup.
// (QW: state -> weight, a callable so that the weights may come from K.q_w or from a copy in LDS WITHOUT a pointer select: a
//  pointer that may be either is a generic pointer, its loads are FLAT loads, and one pending FLAT load anywhere on a path makes
//  every later wait of the kernel a wait for ALL memory operations in flight - vmcnt(0) lgkmcnt(0).)
template <int NTH, class QW>
static __device__ __forceinline__ void fill_coupling_coef_q(double* coefL, double* coefR, const FteConst& K, int node_i,
                                                            int tid, QW qw) {
  const int loc_i = 3 * (node_i - K.pin_left);                 // local index of the node's first frame
  const int64_t f_i = K.n_offset + (int64_t)loc_i;
  for (int e = tid; e < 2 * 9 * NP; e += NTH) {
    const int side = e / (9 * NP), q = e % (9 * NP), pair = q / NP, p = q % NP, ii = pair / 3, jj = pair % 3;
    if (side == 0 ? !coefL : !coefR) continue;                 // (a caller that only wants one side)
    double v = 0.0;
    if (ii <= jj) {
      const int k = 3 + ii - jj;
      // left table: frame jj of node i-1 with frame ii of node i; right table: frame jj of node i with frame ii of node i+1.
      // A slot beyond the local frames (the last node of a window that ends inside the sequence holds 1 or 2 live
      // frames) is an identity row of the chain and must not be coupled, whatever the global band says there.
      const int hi = side == 0 ? loc_i + ii : loc_i + 3 + ii;
      if (hi < K.n_frames)
        v = 2.0 * qw(p) * (side == 0 ? band_coef_clip(f_i - 3 + jj, k, K.n_global, K.clip_len)
                                             : band_coef_clip(f_i + jj, k, K.n_global, K.clip_len));
    }
    if (side == 0) coefL[q] = v;
    else coefR[q] = v;
  }
}
template <int NTH = 256>
static __device__ __forceinline__ void fill_coupling_coef(double* coefL, double* coefR, const FteConst& K, int node_i, int tid) {
  fill_coupling_coef_q<NTH>(coefL, coefR, K, node_i, tid, [&](int p) { return K.q_w[p]; });
}
template <int NTH = 256>
static __device__ __forceinline__ void fill_coupling_coef(double* coefL, double* coefR, const FteConst& K, int node_i, int tid,
                                                          const double* qw) {
  fill_coupling_coef_q<NTH>(coefL, coefR, K, node_i, tid, [&](int p) { return qw[p]; });
}

// True when the coupling tables of every node first .. last are those of an interior node of one long sequence - no clip
// structure, every frame slot alive, every third-difference row involved at least three frames away from either end of the
// sequence (band_coef's interior branch) - i.e. identical: a kernel walking a run of nodes fills them once.  (Conservative
// by a node on either side.)
static __device__ __forceinline__ bool coupling_tables_uniform(const FteConst& K, int first, int last) {
  if (K.clip_len > 0) return false;
  const int64_t lo = K.n_offset + 3 * (int64_t)(first - K.pin_left) - 3;   // first frame a left table of `first` refers to
  const int64_t hi = K.n_offset + 3 * (int64_t)(last - K.pin_left) + 5;    // last frame a right table of `last` refers to
  return lo >= 3 && hi + 3 <= K.n_global - 4 && 3 * (last - K.pin_left) + 5 < K.n_frames;
}

// True when the RIGHT table of one node is that of an interior node (every frame of the node and of its right neighbour alive, in
// one clip, no third-difference row cut by an end of the sequence / clip): the entry (ii, jj), ii <= jj, is then
// 2 q_p {-1, 6, -15}[jj - ii] - what right_table_interior writes.
static __device__ __forceinline__ bool right_table_is_interior(const FteConst& K, int node) {
  const int loc = 3 * (node - K.pin_left);
  const int64_t f = K.n_offset + (int64_t)loc;
  const int64_t len = K.clip_len > 0 ? K.clip_len : K.n_global;
  const int64_t r = K.clip_len > 0 ? f % K.clip_len : f;
  return f >= 0 && r <= len - 6 && f + 5 < K.n_global && loc + 5 < K.n_frames;
}
template <int NTH, class QW>
static __device__ __forceinline__ void right_table_interior(double* coefR, int tid, QW qw) {
  for (int q = tid; q < 9 * NP; q += NTH) {
    const int pair = q / NP, p = q % NP, ii = pair / 3, jj = pair % 3;
    coefR[q] = ii <= jj ? 2.0 * qw(p) * (jj == ii ? -1.0 : (jj - ii == 1 ? 6.0 : -15.0)) : 0.0;
  }
}

// Chains with fused narrow levels (seplevel.hip) keep y = U^T b of eliminated nodes in an array of its own.
__device__ __forceinline__ double* ybuf(const BcrChain& ch) { return ch.Y ? ch.Y : ch.b; }

// (fused narrow levels, seplevel.hip; the isolated level inside k_sep_tail)
// D_i + AL_i + SL_i + SR_i (lower tiles; the sums only inside the 75 x 75 live part) -> Lm, b_i + their rows 79 -> yv.
// Absent terms are read from D itself and masked afterwards (a conditional load compiles to a masked load with a full wait per
// element: bcr.hip, k_bcr_update_deep).  All loads of the thread are in flight before the first LDS write.
template <int NTH>
__device__ __forceinline__ void load_node_sum(double* Lm, double* yv, const BcrChain& ch, int i, int fl, int tid) {
  constexpr int NQ = (LOWER_ITEMS + NTH - 1) / NTH;
  const size_t MB = (size_t)BS * BS;
  const double* Dg = ch.D + i * MB;
  const bool hal = ch.AL0 != nullptr, hsl = (fl & 1) != 0, hsr = (fl & 2) != 0;
  const double* Ag = hal ? ch.AL0 + i * MB : Dg;
  const double* Lg = hsl ? ch.SL + i * MB : Dg;
  const double* Rg = hsr ? ch.SR + i * MB : Dg;
  double2 dv[NQ], av[NQ], lv[NQ], rv[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int idx = tid + NTH * k;
    if (idx < LOWER_ITEMS) {
      int rr, cc;
      lower_item(idx, rr, cc);
      dv[k] = *reinterpret_cast<const double2*>(Dg + rr * BS + cc);
      av[k] = *reinterpret_cast<const double2*>(Ag + rr * BS + cc);
      lv[k] = *reinterpret_cast<const double2*>(Lg + rr * BS + cc);
      rv[k] = *reinterpret_cast<const double2*>(Rg + rr * BS + cc);
    }
  }
  double bb = 0.0, ab = 0.0, lb = 0.0, rb = 0.0;
  if (tid < BS) {
    bb = ch.b[(size_t)i * BS + tid];
    const int c = tid < 3 * NP ? tid : 0;
    ab = Ag[(size_t)(BS - 1) * BS + c];
    lb = Lg[(size_t)(BS - 1) * BS + c];
    rb = Rg[(size_t)(BS - 1) * BS + c];
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int idx = tid + NTH * k;
    if (idx < LOWER_ITEMS) {
      int rr, cc;
      lower_item(idx, rr, cc);
      const bool in0 = rr < 3 * NP && cc < 3 * NP, in1 = rr < 3 * NP && cc + 1 < 3 * NP;
      double x = dv[k].x, y = dv[k].y;
      x += (hal && in0) ? av[k].x : 0.0;
      y += (hal && in1) ? av[k].y : 0.0;
      x += (hsl && in0) ? lv[k].x : 0.0;
      y += (hsl && in1) ? lv[k].y : 0.0;
      x += (hsr && in0) ? rv[k].x : 0.0;
      y += (hsr && in1) ? rv[k].y : 0.0;
      Lm[rr * LD + cc] = x;
      Lm[rr * LD + cc + 1] = y;
    }
  }
  if (tid < BS) {
    const bool live = tid < 3 * NP;
    double v = bb;
    v += (hal && live) ? ab : 0.0;
    v += (hsl && live) ? lb : 0.0;
    v += (hsr && live) ? rb : 0.0;
    yv[tid] = v;
  }
}


}  // namespace acino
