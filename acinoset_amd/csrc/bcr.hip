// Block cyclic reduction (BCR) for the block-tridiagonal Gauss-Newton system of the FTE solve.
//
// Unknowns are grouped in super-blocks ("nodes") of 3 frames x 25 states = 75, padded to 80 = 5 tiles
// of 16 with identity rows, so every block operation is a 5x5 grid of 16x16 fp64 tiles executed on
// the matrix cores (v_mfma_f64_16x16x4_f64).  One level = two launches:
//   elim   (one workgroup per eliminated node i with neighbours l, r):
//            D_i = L L^T (blocked Cholesky in LDS) with U = L^-T built alongside ([A; I] L^-T = [L; L^-T]),
//            W_l = U^T A_il, W_r = U^T A_ir (plain tile GEMMs), y = U^T b_i         -> HBM
//   update (three workgroups per remaining node j, one per line):
//            D_j -= W_r(i-)^T W_r(i-) + W_l(i+)^T W_l(i+)
//            new coupling block(j', j) = -W_r(i+)^T W_l(i+)
//            b_j -= W_r(i-)^T y(i-) + W_l(i+)^T y(i+)          (a mat-vec straight from HBM, no LDS staging)
// and back-substitution x_i = U (y - W_l x_l - W_r x_r) (three mat-vecs) walks the levels in reverse.
// Level-0 couplings are the constant third-difference blocks and are generated in LDS, never stored.
// LDS: ONE 80x81 fp64 matrix per workgroup (52 KB; 2 workgroups per CU in the elimination - 255 VGPRs -, 3 in the
// update); narrow levels have their own latency-oriented kernels (*_deep, backsub_tail); leading dimension 81 makes both
// the row-pattern and the column-pattern MFMA operand reads bank-conflict free.  The damped system itself is
// built inside the level-0 kernels from the assembly's H/g (no separate set-up pass through HBM).
#include "bcr.hpp"
#include "dense80.hpp"
#include "bcr_dev.hpp"
#include "seplevel.hpp"

namespace acino {

__device__ void sparse_coupling_w(const double* Um, double* __restrict__ Wout, const double* coef, bool left, int tid) {
  for (int e = tid; e < BS * BS; e += 256) {
    const int r = e / BS, c = e % BS;
    double v = 0.0;
    if (c < 3 * NP) {
      const int cj = c / NP, p = c % NP;
      if (left) {
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
          const int row = ii * NP + p;
          if (ii <= cj && row <= r) v += Um[row * LD + r] * coef[(ii * 3 + cj) * NP + p];
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const int row = jj * NP + p;
          if (jj >= cj && row <= r) v += Um[row * LD + r] * coef[(cj * 3 + jj) * NP + p];
        }
      }
    }
    Wout[e] = v;
  }
}

// W(:, strip) = U^T A(:, strip) with the B operand read straight from HBM/L2 ONCE per strip: lane (li, lk)
// holds A[4t+lk][cc+li] for t = 0..19 (element A[row][col] at Ag[row*rs + col*cs]; rs/cs select the plain or
// the transposed coupling block) and every row tile IB reuses the first 4(IB+1) of them:
//   W(IB, strip) = sum_{k<=IB} X(IB,k) A(k, strip),   X(IB,k)[i][kk] = U[k16+kk][IB16+i]  (A operand, from LDS).
template <int IB>
__device__ __forceinline__ void strip_row_r(const double* Lm, const double (&bv)[20], double* __restrict__ Wg, int cc,
                                            int li, int lk) {
  constexpr int NS = 4 * (IB + 1);
  double av[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) av[t] = Lm[(4 * t + lk) * LD + IB * 16 + li];
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < NS; ++t) acc = mfma(av[t], bv[t], acc);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Wg[(IB * 16 + lk + 4 * rr) * BS + cc + li] = acc[rr];
}
__device__ __forceinline__ void gemm_strip_g(const double* Lm, const double* __restrict__ Ag, int rs, int cs,
                                             double* __restrict__ Wg, int cc, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  double bv[20];
#pragma unroll
  for (int t = 0; t < 20; ++t) bv[t] = Ag[(4 * t + lk) * rs + (cc + li) * cs];
  strip_row_r<4>(Lm, bv, Wg, cc, li, lk);
  strip_row_r<3>(Lm, bv, Wg, cc, li, lk);
  strip_row_r<2>(Lm, bv, Wg, cc, li, lk);
  strip_row_r<1>(Lm, bv, Wg, cc, li, lk);
  strip_row_r<0>(Lm, bv, Wg, cc, li, lk);
}

// Eliminate node i: D_i = L L^T, U = L^-T, W_l = U^T A_il, W_r = U^T A_ir, y = U^T b_i.  Stores U (in the D
// slot), W_l, W_r and y.  LDS holds ONE 80x81 matrix (the factor), so two workgroups share a CU: the serial
// pivot chains of one overlap the matrix-core / memory phases of the others.
__global__ void __launch_bounds__(256, 2)   // <= 256 VGPR+AGPR: two workgroups per CU overlap each other's pivot chains
k_bcr_elim(BcrChain ch, const int* __restrict__ elim, const FteConst* __restrict__ cst, int* numeric_err,
           const int* __restrict__ status, int level) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Lm = reinterpret_cast<double*>(smem_raw);
  double* yv = Lm + MAT;       // [80] rhs
  double* red = yv + BS;       // [8]
  double* coefL = red + 8;     // [225]
  double* coefR = coefL + 9 * NP;
  double* ysc = coefR + 9 * NP;   // [3][80] partial sums of the triangular mat-vecs
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int blk = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // one contiguous range of the chain per XCD
  const int i = elim[3 * blk], l = elim[3 * blk + 1], r = elim[3 * blk + 2];
  const size_t MB = (size_t)BS * BS;
  const bool fused = ch.st != nullptr && level == 0;
  const bool impl_l = ch.implicit_couplings && l >= 0 && l == i - 1, impl_r = ch.implicit_couplings && r >= 0 && r == i + 1;
  if (!fused && (impl_l || impl_r)) fill_coupling_coef(coefL, coefR, *cst, i, tid);   // visible after the barriers below
// (debug stamps: workgroup dbg[64] of the launch at level dbg[65] writes wall-clock ticks of its phases into dbg[0..63])
#define ACINO_STAMP(k) do { if (ch.dbg && tid == 0 && (long long)blockIdx.x == ch.dbg[64] && (long long)level == ch.dbg[65]) ch.dbg[k] = (long long)wall_clock64(); } while (0)
  ACINO_STAMP(0);
  if (fused) {
    double gmax = build_node(Lm, yv, ch, *cst, i, tid);
    publish_gmax(gmax, red, ch.gn_part, i, tid);
  } else {
    load_mat(Lm, ch.D + i * MB, tid);      // (only the lower tiles are meaningful: the update kernels write no others)
    if (tid < BS) yv[tid] = ch.b[(size_t)i * BS + tid];
  }
  __syncthreads();
  ACINO_STAMP(1);
  chol80(Lm, tid, numeric_err, (ch.dbg && (long long)blockIdx.x == ch.dbg[64] && (long long)level == ch.dbg[65]) ? ch.dbg : nullptr);
  ACINO_STAMP(2);
  if (fused) ACINO_STAMP(3);
  if (fused) {
    // Level 0 of an FTE chain: both couplings are the sparse third-difference blocks E, so the consumers
    // need only G = D_i^-1 = U U^T (W^T W = E^T G E is a <= 9-term stencil) and z = D_i^-1 b = U y.
    // y = U^T b, then z = U y: three partial sums per row (240 threads), so a row costs <= 27 dependent FMAs
    const int row = tid % BS, part = tid / BS;
    if (tid < 3 * BS) {
      double yy = 0.0;
      const int c1 = min(27 * part + 27, row + 1);
      for (int c = 27 * part; c < c1; ++c) yy += Lm[c * LD + row] * yv[c];
      ysc[tid] = yy;
    }
    __syncthreads();
    if (tid < BS) yv[tid] = ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
    __syncthreads();
    if (tid < 3 * BS) {
      double z = 0.0;
      const int c1 = min(27 * part + 27, BS);
      for (int c = max(27 * part, row); c < c1; ++c) z += Lm[row * LD + c] * yv[c];
      ysc[tid] = z;
    }
    __syncthreads();
    if (tid < BS) ch.b[(size_t)i * BS + tid] = ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
    double* Gg = ch.D + i * MB;
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = wave + 4 * q;
      if (t < 15) {
        const int ib = tri_i(t), jb = tri_j(t);
        const double* pa = Lm + (ib * 16 + li) * LD + ib * 16 + lk;   // U(ib, k >= ib)[i][kk]
        const double* pb = Lm + (jb * 16 + li) * LD + ib * 16 + lk;   // U(jb, k >= ib)[j][kk]
        d4 acc = {0, 0, 0, 0};
        switch (ib) {
          case 0: acc = mma_seq<20, false>(acc, pa, 4, pb, 4); break;
          case 1: acc = mma_seq<16, false>(acc, pa, 4, pb, 4); break;
          case 2: acc = mma_seq<12, false>(acc, pa, 4, pb, 4); break;
          case 3: acc = mma_seq<8, false>(acc, pa, 4, pb, 4); break;
          default: acc = mma_seq<4, false>(acc, pa, 4, pb, 4); break;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          Gg[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = acc[rr];
          if (ib != jb) Gg[(jb * 16 + li) * BS + ib * 16 + lk + 4 * rr] = acc[rr];
        }
      }
    }
    ACINO_STAMP(4);
    ACINO_STAMP(5);
    return;
  }
  if (l >= 0) {
    if (impl_l) {
      sparse_coupling_w(Lm, ch.Wl + i * MB, coefL, true, tid);
    } else {
      const double* A = ch.Cpl + l * MB;                  // block(i, l): rows i, cols l
      for (int ct = wave; ct < 5; ct += 4) gemm_strip_g(Lm, A, BS, 1, ch.Wl + i * MB, ct * 16, lane);
    }
  }
  if (!fused) ACINO_STAMP(3);       // (wave 0: after its W_l strips, before its W_r strip)
  if (r >= 0) {
    if (impl_r) {
      sparse_coupling_w(Lm, ch.Wr + i * MB, coefR, false, tid);
    } else {
      const double* A = ch.Cpl + i * MB;                  // block(r, i)^T: rows i, cols r
      for (int ct = 3 - wave; ct < 5; ct += 4) gemm_strip_g(Lm, A, 1, BS, ch.Wr + i * MB, ct * 16, lane);
    }
  }
  if (tid < 3 * BS) {                                                  // y = U^T b in three partial sums per row
    const int row = tid % BS, part = tid / BS;
    double yy = 0.0;
    const int c1 = min(27 * part + 27, row + 1);
    for (int c = 27 * part; c < c1; ++c) yy += Lm[c * LD + row] * yv[c];
    ysc[tid] = yy;
  }
  ACINO_STAMP(4);
  store_mat(ch.U + i * MB, Lm, tid);
  __syncthreads();
  if (tid < BS) ybuf(ch)[(size_t)i * BS + tid] = ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
  ACINO_STAMP(5);
}

// Elimination at the NARROW levels (explicit couplings only).  The ten 16-column strips of [W_l | W_r] of one node
// are spread over T workgroups, each of which repeats the factorisation (the chip is idle: redundant work is free,
// the pivot chain is not) and then computes only its strips - row tiles spread over the four waves - from B operands
// that were requested from HBM BEFORE the factorisation.  Workgroup 0 of a node also stores U and y.
// grid = n_elim * T;  strips s = g, g + T, ... < 10;  s < 5: W_l strip s, else W_r strip s - 5.
__device__ __forceinline__ void deep_strip_operands(const BcrChain& ch, int i, int l, int r, int sidx, double (&bv)[20],
                                                    int li, int lk) {
  const size_t MB = (size_t)BS * BS;
  const bool left = sidx < 5;
  const int cc = 16 * (left ? sidx : sidx - 5);
  const int nb = left ? l : r;
  if (nb < 0) return;
  // W_l: block(i, l) stored at Cpl[l], element [row i][col l] ; W_r: block(r, i)^T stored at Cpl[i], [row r][col i]
  const double* Ag = ch.Cpl + (left ? (size_t)l : (size_t)i) * MB;
  const int rs = left ? BS : 1, cs = left ? 1 : BS;
#pragma unroll
  for (int t = 0; t < 20; ++t) bv[t] = Ag[(4 * t + lk) * rs + (cc + li) * cs];
}
template <int IB>
__device__ __forceinline__ void deep_row_tile(const double* Lm, const double (&bv)[20], double* __restrict__ Wg, int cc,
                                              int li, int lk) {
  strip_row_r<IB>(Lm, bv, Wg, cc, li, lk);
}
__global__ void __launch_bounds__(256, 2)
k_bcr_elim_deep(BcrChain ch, const int* __restrict__ elim, int* numeric_err, const int* __restrict__ status, int T,
                int extra, int nx, int per, int total, const double* __restrict__ AL) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Lm = reinterpret_cast<double*>(smem_raw);
  double* yv = Lm + MAT;       // [80] rhs
  double* ysc = yv + BS;       // [3][80]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  // narrow level on nx XCDs (workgroup k runs on XCD k % 8), each a contiguous range of `per` logical workgroups:
  // the workgroups of a node and of its neighbours share one L2, and so do this kernel and its consumers' kernels
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bx = xcd * per + slot;
  if (xcd >= nx || slot >= per || bx >= total) return;
  // T (+ 1) workgroups per node: T share the ten strips of W_l / W_r; where the level leaves CUs free (extra), one more
  // stores the factor and y, so that no workgroup has a strip AND the 51 KB store behind its factorisation
  const int ent = bx / (T + extra), g = bx % (T + extra), g_store = extra ? T : 0;
  const int i = elim[3 * ent], l = elim[3 * ent + 1], r = elim[3 * ent + 2];
  const size_t MB = (size_t)BS * BS;
  double bv[20];
  if (g < T) deep_strip_operands(ch, i, l, r, g, bv, li, lk);       // first strip: in flight during the factorisation
  // level 0 of the chunked solver's separator chain: D_i and b_i still lack the contribution AL of the run on the node's
  // right (lower tiles, rows / columns < 75; row 79 = its update of b) - added here instead of by a launch of its own
  constexpr int NQA = (LOWER_ITEMS + 255) / 256;
  double2 av[NQA];
  double ab = 0.0;
  if (AL) {
    const double* A = AL + i * MB;
#pragma unroll
    for (int k = 0; k < NQA; ++k) {
      const int idx = tid + 256 * k;
      if (idx < LOWER_ITEMS) {
        int rr, cc;
        lower_item(idx, rr, cc);
        av[k] = *reinterpret_cast<const double2*>(A + rr * BS + cc);
      }
    }
    if (tid < 3 * NP) ab = A[(size_t)(BS - 1) * BS + tid];
  }
  load_mat(Lm, ch.D + i * MB, tid);
  if (tid < BS) yv[tid] = ch.b[(size_t)i * BS + tid] + ab;
  __syncthreads();
  if (AL) {
#pragma unroll
    for (int k = 0; k < NQA; ++k) {
      const int idx = tid + 256 * k;
      if (idx < LOWER_ITEMS) {
        int rr, cc;
        lower_item(idx, rr, cc);
        if (rr < 3 * NP) {
          if (cc < 3 * NP) Lm[rr * LD + cc] += av[k].x;
          if (cc + 1 < 3 * NP) Lm[rr * LD + cc + 1] += av[k].y;
        }
      }
    }
    __syncthreads();
  }
  chol80(Lm, tid, g == g_store ? numeric_err : nullptr);
  for (int sidx = g; sidx < 10 && g < T; sidx += T) {
    if (sidx != g) deep_strip_operands(ch, i, l, r, sidx, bv, li, lk);
    const bool left = sidx < 5;
    if ((left ? l : r) < 0) continue;
    double* Wg = (left ? ch.Wl : ch.Wr) + i * MB;
    const int cc = 16 * (left ? sidx : sidx - 5);
    // row tiles of the strip over the waves: {4}, {3}, {2, 0}, {1}  (20, 16, 16, 8 matrix-core steps)
    if (wave == 0) deep_row_tile<4>(Lm, bv, Wg, cc, li, lk);
    else if (wave == 1) deep_row_tile<3>(Lm, bv, Wg, cc, li, lk);
    else if (wave == 2) {
      deep_row_tile<2>(Lm, bv, Wg, cc, li, lk);
      deep_row_tile<0>(Lm, bv, Wg, cc, li, lk);
    } else deep_row_tile<1>(Lm, bv, Wg, cc, li, lk);
  }
  if (g == g_store) {
    if (tid < 3 * BS) {                                                  // y = U^T b in three partial sums per row
      const int row = tid % BS, part = tid / BS;
      double yy = 0.0;
      const int c1 = min(27 * part + 27, row + 1);
      for (int c = 27 * part; c < c1; ++c) yy += Lm[c * LD + row] * yv[c];
      ysc[tid] = yy;
    }
    // (into ch.U, NOT over D_i: the sibling workgroups of this node may not have loaded D_i yet - on a busy GPU they are
    //  dispatched late - and would factor a mixture of D_i and U.  Found in round 2 with several contexts running
    //  concurrently; alone, all siblings start together and the in-place store went unnoticed.)
    store_mat(ch.U + i * MB, Lm, tid);
    __syncthreads();
    if (tid < BS) ybuf(ch)[(size_t)i * BS + tid] = ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
  }
}

// Half-height (40-row) staging of an 80x80 matrix: rows [r0, r0+40) -> dst[40][LD].
__device__ __forceinline__ void load_half(double* dst, const double* __restrict__ src, int r0, int tid) {
  const double2* s2 = reinterpret_cast<const double2*>(src + (size_t)r0 * BS);
  double2 v[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int idx = tid + 256 * k;
    if (idx < 40 * BS / 2) v[k] = s2[idx];
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int idx = tid + 256 * k;
    if (idx < 40 * BS / 2) {
      const int e = 2 * idx, r = e / BS, c = e % BS;
      dst[r * LD + c] = v[k].x;
      dst[r * LD + c + 1] = v[k].y;
    }
  }
}

// role 0: D_j -= W_r(i-)^T W_r(i-) + W_l(i+)^T W_l(i+), b_j -= W^T y ; role 1: block(jn, j) = -W_r(i+)^T W_l(i+).
// Both roles stream through ONE 80x81 LDS buffer (52 KB) so three workgroups share a CU and the staging of
// one overlaps the matrix-core phase of another.  The matrix-core phase is LDS-bandwidth bound when every MFMA
// fetches its own A and B operand, so tiles are assigned to waves BY ROW and the k loop is outermost: the operand
// of a k-step depends only on (k, column block), one LDS read feeds every tile of the row that uses the block
// (role 0: 14 reads per k-step for 15 tiles instead of 30; role 1: 28 for 25 tiles instead of 50).
//
// role 0 tiles (ib, jb), jb <= ib:  wave 0: row 4 | wave 1: row 3 | wave 2: row 2 and (0,0) | wave 3: row 1
template <int WAVE>
struct SyrkTiles {
  static constexpr int nb = WAVE == 0 ? 5 : (WAVE == 1 ? 4 : (WAVE == 2 ? 3 : 2));   // column blocks 0 .. nb-1
  static constexpr int nt = WAVE == 0 ? 5 : (WAVE == 1 ? 4 : (WAVE == 2 ? 4 : 2));
  static constexpr int ib(int q) { return WAVE == 0 ? 4 : (WAVE == 1 ? 3 : (WAVE == 2 ? (q < 3 ? 2 : 0) : 1)); }
  static constexpr int jb(int q) { return WAVE == 2 ? (q < 3 ? q : 0) : q; }
};
template <int WAVE, int KSTEPS>
__device__ __forceinline__ void syrk_rows(const double* Wb, d4 (&acc)[5], int li, int lk) {
  using T = SyrkTiles<WAVE>;
  const double* p = Wb + lk * LD + li;
  double v0[T::nb], v1[T::nb];
#pragma unroll
  for (int c = 0; c < T::nb; ++c) v0[c] = p[c * 16];
#pragma unroll
  for (int s = 0; s < KSTEPS; s += 2) {
#pragma unroll
    for (int c = 0; c < T::nb; ++c) v1[c] = p[(4 * (s + 1)) * LD + c * 16];
#pragma unroll
    for (int q = 0; q < T::nt; ++q) acc[q] = mfma(-v0[T::ib(q)], v0[T::jb(q)], acc[q]);
    if (s + 2 < KSTEPS) {
#pragma unroll
      for (int c = 0; c < T::nb; ++c) v0[c] = p[(4 * (s + 2)) * LD + c * 16];
    }
#pragma unroll
    for (int q = 0; q < T::nt; ++q) acc[q] = mfma(-v1[T::ib(q)], v1[T::jb(q)], acc[q]);
  }
}
template <int WAVE, class F>
__device__ __forceinline__ void syrk_tiles_foreach(F&& f) {
  using T = SyrkTiles<WAVE>;
#pragma unroll
  for (int q = 0; q < T::nt; ++q) f(q, T::ib(q), T::jb(q));
}
// role 1 tiles (ib, jb), all 25:  wave w: row w (5 tiles) + from row 4: (4,0) | (4,1) | (4,2) | (4,3),(4,4)
template <int WAVE>
struct GemmTiles {
  static constexpr int nt = WAVE == 3 ? 7 : 6;
  static constexpr int ib(int q) { return q < 5 ? WAVE : 4; }
  static constexpr int jb(int q) { return q < 5 ? q : (WAVE == 3 ? 3 + (q - 5) : WAVE); }
};
template <int WAVE, int KSTEPS>
__device__ __forceinline__ void gemm_rows(const double* Ha, const double* Hb, d4 (&acc)[7], int li, int lk) {
  using T = GemmTiles<WAVE>;
  const double* pa = Ha + lk * LD + li;
  const double* pb = Hb + lk * LD + li;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    double a0 = pa[(4 * s) * LD + WAVE * 16], a4 = pa[(4 * s) * LD + 4 * 16], bq[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) bq[c] = pb[(4 * s) * LD + c * 16];
#pragma unroll
    for (int q = 0; q < T::nt; ++q) acc[q] = mfma(q < 5 ? -a0 : -a4, bq[T::jb(q)], acc[q]);
  }
}

__global__ void __launch_bounds__(256, 3)
k_bcr_update(BcrChain ch, const int* __restrict__ remain, const FteConst* __restrict__ cst,
             const int* __restrict__ status, int level) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Wb = reinterpret_cast<double*>(smem_raw);
  double* yv = Wb + MAT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int blk = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // the roles of a node and its neighbours: one L2
  const int ent = blk / 3, role = blk % 3;
  const int j = remain[4 * ent], im = remain[4 * ent + 1], ip = remain[4 * ent + 2], jn = remain[4 * ent + 3];
  const size_t MB = (size_t)BS * BS;
  if (role == 2) {   // b_j -= W_r(im)^T y(im) + W_l(ip)^T y(ip), straight from HBM (see k_bcr_update_deep)
    if (im < 0 && ip < 0) return;
    double* ya = Wb;            // [80] y(im), [80] y(ip), [3][80] partial sums
    double* yb = ya + BS;
    double* ysc = yb + BS;
    if (tid < BS) {
      ya[tid] = im >= 0 ? ybuf(ch)[(size_t)im * BS + tid] : 0.0;
      yb[tid] = ip >= 0 ? ybuf(ch)[(size_t)ip * BS + tid] : 0.0;
    }
    const int col = tid % BS, part = tid / BS, k0 = 27 * part, nk = part < 2 ? 27 : 26;
    double wa[27], wc[27];
    if (tid < 3 * BS) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        wa[k] = (im >= 0 && k < nk) ? ch.Wr[im * MB + (size_t)(k0 + k) * BS + col] : 0.0;
        wc[k] = (ip >= 0 && k < nk) ? ch.Wl[ip * MB + (size_t)(k0 + k) * BS + col] : 0.0;
      }
    }
    __syncthreads();
    if (tid < 3 * BS) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        s0 += wa[k] * ya[k0 + (k < nk ? k : 0)];
        s1 += wc[k] * yb[k0 + (k < nk ? k : 0)];
      }
      ysc[tid] = s0 + s1;
    }
    __syncthreads();
    if (tid < BS) ch.b[(size_t)j * BS + tid] -= ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid];
    return;
  }
  if (role == 0) {
    if (im < 0 && ip < 0) return;
    double* Dj = ch.D + j * MB;
    d4 acc[5];
    auto load_tiles = [&](int q, int ib, int jb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[q][rr] = Dj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li];
    };
    auto store_tiles = [&](int q, int ib, int jb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Dj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = acc[q][rr];   // lower tiles only
    };
    if (wave == 0) syrk_tiles_foreach<0>(load_tiles);
    else if (wave == 1) syrk_tiles_foreach<1>(load_tiles);
    else if (wave == 2) syrk_tiles_foreach<2>(load_tiles);
    else syrk_tiles_foreach<3>(load_tiles);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int nb = side == 0 ? im : ip;
      if (nb < 0) continue;
      if (side == 1 && im >= 0) __syncthreads();               // everyone done with the previous W
      load_mat(Wb, (side == 0 ? ch.Wr : ch.Wl) + nb * MB, tid);   // W_r of the left / W_l of the right neighbour
      __syncthreads();
      if (wave == 0) syrk_rows<0, BS / 4>(Wb, acc, li, lk);
      else if (wave == 1) syrk_rows<1, BS / 4>(Wb, acc, li, lk);
      else if (wave == 2) syrk_rows<2, BS / 4>(Wb, acc, li, lk);
      else syrk_rows<3, BS / 4>(Wb, acc, li, lk);
    }
    if (wave == 0) syrk_tiles_foreach<0>(store_tiles);
    else if (wave == 1) syrk_tiles_foreach<1>(store_tiles);
    else if (wave == 2) syrk_tiles_foreach<2>(store_tiles);
    else syrk_tiles_foreach<3>(store_tiles);
  } else {
    if (ip < 0 || jn < 0) return;
    double* Ha = Wb;                  // rows [h*40, h*40+40) of W_r(ip)   (cols = jn)
    double* Hb = Wb + 40 * LD;        // same rows of W_l(ip)             (cols = j)
    d4 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[q] = d4{0, 0, 0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1) __syncthreads();
      load_half(Ha, ch.Wr + ip * MB, 40 * h, tid);
      load_half(Hb, ch.Wl + ip * MB, 40 * h, tid);
      __syncthreads();
      if (wave == 0) gemm_rows<0, 10>(Ha, Hb, acc, li, lk);
      else if (wave == 1) gemm_rows<1, 10>(Ha, Hb, acc, li, lk);
      else if (wave == 2) gemm_rows<2, 10>(Ha, Hb, acc, li, lk);
      else gemm_rows<3, 10>(Ha, Hb, acc, li, lk);
    }
    double* Cj = ch.Cpl + j * MB;     // block(jn, j): rows jn, cols j
    auto store_c = [&](int q, int ib, int jb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Cj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = acc[q][rr];
    };
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      if (wave == 0 && q < GemmTiles<0>::nt) store_c(q, GemmTiles<0>::ib(q), GemmTiles<0>::jb(q));
      if (wave == 1 && q < GemmTiles<1>::nt) store_c(q, GemmTiles<1>::ib(q), GemmTiles<1>::jb(q));
      if (wave == 2 && q < GemmTiles<2>::nt) store_c(q, GemmTiles<2>::ib(q), GemmTiles<2>::jb(q));
      if (wave == 3 && q < GemmTiles<3>::nt) store_c(q, GemmTiles<3>::ib(q), GemmTiles<3>::jb(q));
    }
  }
}

// The same update for the NARROW levels of the reduction (a handful of nodes: the chip is almost idle and every
// kernel is a latency chain).  One node's work is spread over 2 S workgroups - role 0 / role 1 as above, each split S
// ways by output tile - so the matrix-core phase of a workgroup is one or two tiles per wave instead of four to
// seven (that phase is LDS-bandwidth bound inside one CU); both neighbour matrices are requested at once into two
// LDS buffers, the D_j tile is requested with them and only added at the end.  grid = n_remain * 2 * S.
__global__ void __launch_bounds__(256)
k_bcr_update_deep(BcrChain ch, const int* __restrict__ remain, const int* __restrict__ status, int S, int nx, int per,
                  int total, const double* __restrict__ AL) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Wb = reinterpret_cast<double*>(smem_raw);
  double* Wb2 = Wb + MAT;
  double* yv = Wb2 + MAT;
  double* yv2 = yv + BS;
  double* ysc = yv2 + BS;            // [3][80]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bx = xcd * per + slot;     // (as in k_bcr_elim_deep)
  if (xcd >= nx || slot >= per || bx >= total) return;
  // 2 S + 1 workgroups per node: S for role 0 (D_j tiles), S for role 1 (coupling block), one for b_j
  const int ent = bx / (2 * S + 1), rb = bx % (2 * S + 1), role = rb == 2 * S ? 2 : rb / S, sub = rb % S;
  const int j = remain[4 * ent], im = remain[4 * ent + 1], ip = remain[4 * ent + 2], jn = remain[4 * ent + 3];
  const size_t MB = (size_t)BS * BS;
  if (role == 2) {   // b_j -= W_r(im)^T y(im) + W_l(ip)^T y(ip), straight from HBM: thread (column, third of the rows),
                     // lanes along a row of W (coalesced); keeps this mat-vec off the tile workgroups' critical path
    if (im < 0 && ip < 0 && !AL) return;
    if (tid < BS) {
      yv[tid] = im >= 0 ? ybuf(ch)[(size_t)im * BS + tid] : 0.0;
      yv2[tid] = ip >= 0 ? ybuf(ch)[(size_t)ip * BS + tid] : 0.0;
    }
    const int col = tid % BS, part = tid / BS, k0 = 27 * part, nk = part < 2 ? 27 : 26;
    double wa[27], wc[27];
    if (tid < 3 * BS) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        wa[k] = (im >= 0 && k < nk) ? ch.Wr[im * MB + (size_t)(k0 + k) * BS + col] : 0.0;
        wc[k] = (ip >= 0 && k < nk) ? ch.Wl[ip * MB + (size_t)(k0 + k) * BS + col] : 0.0;
      }
    }
    __syncthreads();
    if (tid < 3 * BS) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        s0 += wa[k] * yv[k0 + (k < nk ? k : 0)];
        s1 += wc[k] * yv2[k0 + (k < nk ? k : 0)];
      }
      ysc[tid] = s0 + s1;
    }
    __syncthreads();
    // (AL: row 79 of the right run's contribution is its update of b_j - see k_bcr_elim_deep)
    if (tid < BS)
      ch.b[(size_t)j * BS + tid] = ch.b[(size_t)j * BS + tid] - (ysc[tid] + ysc[BS + tid] + ysc[2 * BS + tid]) +
                                   ((AL && tid < 3 * NP) ? AL[j * MB + (size_t)(BS - 1) * BS + tid] : 0.0);
    return;
  }
  if (role == 0) {
    if (im < 0 && ip < 0 && !AL) return;
    double* Dj = ch.D + j * MB;
    // (the run's contribution: loaded UNCONDITIONALLY beside the tile - from D_j itself when there is none - and masked
    //  afterwards; `if (AL && r < 75 && c < 75) dt += AL[..]` compiled to a masked load with s_waitcnt vmcnt(0) per element:
    //  16 serialised round trips in front of the products the tiles were "requested first" for)
    const double* Aj = AL ? AL + j * MB : Dj;
    d4 dt[4], da[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {    // this wave's output tiles of D_j, requested first, consumed last
      const int t = min(sub + S * (wave + 4 * q), 14);
      const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r_ = ib * 16 + lk + 4 * rr, c_ = jb * 16 + li;
        dt[q][rr] = Dj[r_ * BS + c_];
        da[q][rr] = Aj[r_ * BS + c_];
      }
    }
    auto fold_al = [&]() {                               // (at the point of use: the tiles are "requested first, consumed last")
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = min(sub + S * (wave + 4 * q), 14);
        const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r_ = ib * 16 + lk + 4 * rr, c_ = jb * 16 + li;
          asm volatile("" : "+v"(da[q][rr]));            // (the load stays a load: no sinking into the select)
          dt[q][rr] += (AL && r_ < 3 * NP && c_ < 3 * NP) ? da[q][rr] : 0.0;
        }
      }
    };
    if (im < 0 && ip < 0) {          // (no eliminated neighbour: only the run's contribution is added)
      fold_al();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = sub + S * (wave + 4 * q);
        if (t < 15) {
          const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) Dj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = dt[q][rr];
        }
      }
      return;
    }
    const bool both = im >= 0 && ip >= 0;
    if (both) load_mat2(Wb, ch.Wr + im * MB, Wb2, ch.Wl + ip * MB, tid);
    else load_mat(Wb, im >= 0 ? ch.Wr + im * MB : ch.Wl + ip * MB, tid);
    __syncthreads();
    fold_al();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = sub + S * (wave + 4 * q);
      if (t < 15) {
        const int ib = tri_i(t), jb = tri_j(t);
        d4 a = {0, 0, 0, 0};
        a = mma_seq<BS / 4, true>(a, Wb + lk * LD + ib * 16 + li, 4 * LD, Wb + lk * LD + jb * 16 + li, 4 * LD);
        if (both) a = mma_seq<BS / 4, true>(a, Wb2 + lk * LD + ib * 16 + li, 4 * LD, Wb2 + lk * LD + jb * 16 + li, 4 * LD);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          Dj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = dt[q][rr] + a[rr];          // lower tiles only
        }
      }
    }
  } else {
    if (ip < 0 || jn < 0) return;
    load_mat2(Wb, ch.Wr + ip * MB, Wb2, ch.Wl + ip * MB, tid);    // W_r(ip): cols = jn ; W_l(ip): cols = j
    __syncthreads();
    double* Cj = ch.Cpl + j * MB;     // block(jn, j): rows jn, cols j
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int t = sub + S * (wave + 4 * q);
      if (t < NT * NT) {
        const int ib = t / NT, jb = t % NT;
        d4 a = {0, 0, 0, 0};
        a = mma_seq<BS / 4, true>(a, Wb + lk * LD + ib * 16 + li, 4 * LD, Wb2 + lk * LD + jb * 16 + li, 4 * LD);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cj[(ib * 16 + lk + 4 * rr) * BS + jb * 16 + li] = a[rr];
      }
    }
  }
}

// x_i = U (y_i - W_l x_l - W_r x_r),  U = L^-T.  All three matrices are requested at once (into registers: ONE HBM round
// trip instead of three) and take their turn in the single LDS buffer; the mat-vecs use 240 threads (three partial
// sums per row).
// NTH = 256, or 512 for the short launches of a separator chain (latency-bound mat-vecs: more waves shorten each)
template <int NTH>
__global__ void __launch_bounds__(NTH)
k_bcr_backsub(BcrChain ch, const int* __restrict__ elim, const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Mb = reinterpret_cast<double*>(smem_raw);
  double* xv = Mb + MAT;       // [2][80] x_l, x_r
  double* tv = xv + 2 * BS;    // [80]
  double* ysc = tv + BS;       // [NPART][80]
  constexpr int NPART = NTH / BS, WID = (BS + NPART - 1) / NPART, NV = (BS * BS / 2 + NTH - 1) / NTH;
  const int tid = threadIdx.x;
  const int i = elim[3 * blockIdx.x], l = elim[3 * blockIdx.x + 1], r = elim[3 * blockIdx.x + 2];
  const size_t MB = (size_t)BS * BS;
  double2 vl[NV], vr[NV], vu[NV];
  if (l >= 0) fetch_mat<NTH>(vl, ch.Wl + i * MB, tid);
  if (r >= 0) fetch_mat<NTH>(vr, ch.Wr + i * MB, tid);
  fetch_mat<NTH>(vu, ch.U + i * MB, tid);
  double t = (tid < BS) ? ybuf(ch)[(size_t)i * BS + tid] : 0.0;
  if (tid < BS) {
    xv[tid] = l >= 0 ? ch.b[(size_t)l * BS + tid] : 0.0;
    xv[BS + tid] = r >= 0 ? ch.b[(size_t)r * BS + tid] : 0.0;
  }
  const int row = tid % BS, part = tid / BS, c0 = WID * part, c1 = min(c0 + WID, BS);
  auto row_sum = [&](int r) {     // fixed order
    double v = ysc[r];
#pragma unroll
    for (int q = 1; q < NPART; ++q) v += ysc[q * BS + r];
    return v;
  };
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    if ((side == 0 ? l : r) < 0) continue;
    stage_mat<NTH>(Mb, side == 0 ? vl : vr, tid);
    __syncthreads();
    if (tid < NPART * BS) {                     // row `row` of W, columns [c0, c1) (stride-81 rows: conflict free)
      const double* x = xv + side * BS;
      double s0 = 0.0, s1 = 0.0;
      int c = c0;
      for (; c + 1 < c1; c += 2) {
        s0 += Mb[row * LD + c] * x[c];
        s1 += Mb[row * LD + c + 1] * x[c + 1];
      }
      if (c < c1) s0 += Mb[row * LD + c] * x[c];
      ysc[tid] = s0 + s1;
    }
    __syncthreads();
    if (tid < BS) t -= row_sum(tid);
  }
  stage_mat<NTH>(Mb, vu, tid);
  if (tid < BS) tv[tid] = t;
  __syncthreads();
  if (tid < NPART * BS) {                       // x = U t (U upper triangular): columns >= row only
    double s0 = 0.0;
    for (int c = max(c0, row); c < c1; ++c) s0 += Mb[row * LD + c] * tv[c];
    ysc[tid] = s0;
  }
  __syncthreads();
  if (tid < BS) ch.b[(size_t)i * BS + tid] = row_sum(tid);
}

// The deepest levels of the back-substitution as ONE launch.  Per node the three matrices do not depend on anything
// computed here, only the two neighbour solutions do - so every (level, node) gets its own workgroup, requests
// W_l, W_r and U at once (three LDS buffers), and only then waits for the deeper levels: a chain of
// (acquire, 160 doubles, two short mat-vecs, release) per level instead of a kernel launch with three HBM round
// trips.  Entries are ordered deepest level first and a workgroup only ever waits for entries with a LOWER index, so
// progress does not depend on how many workgroups are resident.  done[0] counts finished entries (zeroed before
// the launch); visibility between CUs / XCDs: agent-scope release after the solution is written, agent-scope
// acquire by one lane + barrier before it is read.
__global__ void __launch_bounds__(256)
k_bcr_backsub_tail(BcrChain ch, const int* __restrict__ status, int* __restrict__ numeric_err) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Ml = reinterpret_cast<double*>(smem_raw);
  double* Mr = Ml + MAT;
  double* Mu = Mr + MAT;
  double* xl = Mu + MAT;          // [80] each
  double* xr = xl + BS;
  double* tv = xr + BS;
  double* part = tv + BS;         // [3][80]
  const int tid = threadIdx.x;
  const int* en = ch.d_tail + 4 * blockIdx.x;
  const int i = en[0], l = en[1], r = en[2], need = en[3];
  const size_t MB = (size_t)BS * BS;
  {  // all three matrices in flight before the first LDS write
    const double2* s0 = reinterpret_cast<const double2*>(ch.Wl + i * MB);
    const double2* s1 = reinterpret_cast<const double2*>(ch.Wr + i * MB);
    const double2* s2 = reinterpret_cast<const double2*>(ch.U + i * MB);
    double2 v0[13], v1[13], v2[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const int idx = tid + 256 * k;
      if (idx < BS * BS / 2) {
        v0[k] = l >= 0 ? s0[idx] : make_double2(0.0, 0.0);
        v1[k] = r >= 0 ? s1[idx] : make_double2(0.0, 0.0);
        v2[k] = s2[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const int idx = tid + 256 * k;
      if (idx < BS * BS / 2) {
        const int e = 2 * idx, rr = e / BS, c = e % BS;
        Ml[rr * LD + c] = v0[k].x;  Ml[rr * LD + c + 1] = v0[k].y;
        Mr[rr * LD + c] = v1[k].x;  Mr[rr * LD + c + 1] = v1[k].y;
        Mu[rr * LD + c] = v2[k].x;  Mu[rr * LD + c + 1] = v2[k].y;
      }
    }
  }
  const double yi = tid < BS ? ybuf(ch)[(size_t)i * BS + tid] : 0.0;     // y_i: written by the reduction, long ago
  if (tid == 0) {   // cheap relaxed polls, ONE acquire (cache invalidation) once the deeper levels are done
    // BOUNDED: a workgroup only waits for lower block indices, which the dispatcher starts first in practice but HIP does
    // not promise; if the wait outlives ~0.5 s of polling (a step is < 1 ms) the kernel gives up, flags the step
    // (numeric_err bit 1 -> LM status 6 -> ACINO_ERR_HIP from the solve) and lets the launch drain instead of hanging
    long long polls = 0;
    while (__hip_atomic_load(ch.d_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if (++polls > (1ll << 23)) {
        if (numeric_err) atomicOr(numeric_err, 2);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (tid < BS) {
    xl[tid] = l >= 0 ? __builtin_nontemporal_load(ch.b + (size_t)l * BS + tid) : 0.0;
    xr[tid] = r >= 0 ? __builtin_nontemporal_load(ch.b + (size_t)r * BS + tid) : 0.0;
  }
  __syncthreads();
  const int row = tid % BS, pr = tid / BS, k0 = 27 * pr, k1 = min(k0 + 27, BS);
  if (tid < 3 * BS) {
    double s = 0.0;
    for (int k = k0; k < k1; ++k) s += Ml[row * LD + k] * xl[k] + Mr[row * LD + k] * xr[k];
    part[tid] = s;
  }
  __syncthreads();
  if (tid < BS) tv[tid] = yi - ((part[tid] + part[BS + tid]) + part[2 * BS + tid]);
  __syncthreads();
  if (tid < 3 * BS) {                                                  // x = U t, U upper triangular
    double s = 0.0;
    for (int k = max(k0, row); k < k1; ++k) s += Mu[row * LD + k] * tv[k];
    part[tid] = s;
  }
  __syncthreads();
  if (tid < BS) ch.b[(size_t)i * BS + tid] = (part[tid] + part[BS + tid]) + part[2 * BS + tid];
  __syncthreads();                // the workgroup's stores are ordered before lane 0's agent-scope release
  if (tid == 0) __hip_atomic_fetch_add(ch.d_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- level 0 of an FTE chain: stencil forms --------------------------------------------------
// coefficient tables (fill_coupling_coef): cL/cR[(ii*3 + jj)*NP + p], ii <= jj.
//   E_l(n) = block(n, n-1): rows (ii,p) of n, cols (jj,p) of n-1, value cL_n[(ii*3+jj)*NP+p]
//   E_r(n) = block(n, n+1): rows (jj,p) of n, cols (ii,p) of n+1, value cR_n[(ii*3+jj)*NP+p]
// One workgroup per remaining node j: builds D_j, b_j from the assembly output, subtracts
//   E_r(im)^T G(im) E_r(im) + E_l(ip)^T G(ip) E_l(ip)   and   E_r(im)^T z(im) + E_l(ip)^T z(ip),
// and writes the new coupling block(jn, j) = -E_r(ip)^T G(ip) E_l(ip).  Each product with the sparse E is
// done as two passes of <= 3 terms: T = G E in place in LDS (item = (row, state p): the three frame
// columns of p), then E^T T (item = (state p, column): the three frame rows of p), accumulated into the
// thread's registers (8 items x 3 rows of D_j) while the neighbour's G streams through the one LDS buffer.
__global__ void __launch_bounds__(256)
k_bcr_update0(BcrChain ch, const int* __restrict__ remain, const FteConst* __restrict__ cst,
              const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Gb = reinterpret_cast<double*>(smem_raw);
  double* yv = Gb + MAT;             // [80] b_j, then z of the neighbour
  double* red = yv + BS;             // [8]
  double* cLm = red + 8;             // coupling coefficients of node im (only cR used) and ip (cL, cR)
  double* cRm = cLm + 9 * NP;
  double* cLp = cRm + 9 * NP;
  double* cRp = cLp + 9 * NP;
  const int tid = threadIdx.x;
  const int blk = xcd_contiguous_rev((int)blockIdx.x, (int)gridDim.x);   // neighbouring nodes share G(i) through one L2; backwards: the G written last by the elimination first
  const int j = remain[4 * blk], im = remain[4 * blk + 1], ip = remain[4 * blk + 2], jn = remain[4 * blk + 3];
  const size_t MB = (size_t)BS * BS;
  const FteConst& K = *cst;
  if (im >= 0) fill_coupling_coef(cLm, cRm, K, im, tid);
  if (ip >= 0) fill_coupling_coef(cLp, cRp, K, ip, tid);
  double gmax = build_node(Gb, yv, ch, K, j, tid);
  publish_gmax(gmax, red, ch.gn_part, j, tid);            // barrier: D_j complete in LDS, tables visible
  // ownership: item q = tid + 256 m (m < 8, q < 25*80): state p = q / 80, column c = q % 80, rows (0..2, p)
  double dv[8][3];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int q = tid + 256 * m;
    if (q < NP * BS) {
      const int p = q / BS, c = q % BS;
#pragma unroll
      for (int a = 0; a < 3; ++a) dv[m][a] = Gb[(a * NP + p) * LD + c];
    }
  }
  double* Dj = ch.D + j * MB;
  for (int e = tid; e < 5 * BS; e += 256) Dj[3 * NP * BS + e] = Gb[(3 * NP + e / BS) * LD + e % BS];   // padding rows
  double bs = (tid < BS) ? yv[tid] : 0.0;
  __syncthreads();
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int nb = side == 0 ? im : ip;
    if (nb < 0) continue;
    const double* cE = side == 0 ? cRm : cLp;              // E_r(im) or E_l(ip): the coupling towards j
    load_mat(Gb, ch.D + nb * MB, tid);                     // G of the eliminated neighbour
    if (tid < BS) yv[tid] = ch.b[(size_t)nb * BS + tid];   // its z
    __syncthreads();
    // pass 1: T = G E (columns of j), in place.  item (row r, state p): T[r][(jb,p)] for jb = 0..2
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int q = tid + 256 * m;
      if (q < BS * NP) {
        const int r = q / NP, p = q % NP;
        const double g0 = Gb[r * LD + p], g1 = Gb[r * LD + NP + p], g2 = Gb[r * LD + 2 * NP + p];
        double t0, t1, t2;
        if (side == 0) {   // E_r: column (jb,p) has rows (j2 >= jb, p), coefficient c[(jb*3+j2)*NP+p]
          t0 = g0 * cE[(0 * 3 + 0) * NP + p] + g1 * cE[(0 * 3 + 1) * NP + p] + g2 * cE[(0 * 3 + 2) * NP + p];
          t1 = g1 * cE[(1 * 3 + 1) * NP + p] + g2 * cE[(1 * 3 + 2) * NP + p];
          t2 = g2 * cE[(2 * 3 + 2) * NP + p];
        } else {           // E_l: column (jb,p) has rows (i2 <= jb, p), coefficient c[(i2*3+jb)*NP+p]
          t0 = g0 * cE[(0 * 3 + 0) * NP + p];
          t1 = g0 * cE[(0 * 3 + 1) * NP + p] + g1 * cE[(1 * 3 + 1) * NP + p];
          t2 = g0 * cE[(0 * 3 + 2) * NP + p] + g1 * cE[(1 * 3 + 2) * NP + p] + g2 * cE[(2 * 3 + 2) * NP + p];
        }
        Gb[r * LD + p] = t0;
        Gb[r * LD + NP + p] = t1;
        Gb[r * LD + 2 * NP + p] = t2;
      }
    }
    __syncthreads();
    // pass 2: D_j -= E^T T ; (right neighbour only) block(jn, j) = -E_r(ip)^T T.  item (state p, column c)
    double* Cj = ch.Cpl + j * MB;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int q = tid + 256 * m;
      if (q < NP * BS) {
        const int p = q / BS, c = q % BS;
        const double t0 = Gb[p * LD + c], t1 = Gb[(NP + p) * LD + c], t2 = Gb[(2 * NP + p) * LD + c];
        if (side == 0) {   // rows (ja,p) of E_r^T: sum over j1 >= ja
          dv[m][0] -= cE[(0 * 3 + 0) * NP + p] * t0 + cE[(0 * 3 + 1) * NP + p] * t1 + cE[(0 * 3 + 2) * NP + p] * t2;
          dv[m][1] -= cE[(1 * 3 + 1) * NP + p] * t1 + cE[(1 * 3 + 2) * NP + p] * t2;
          dv[m][2] -= cE[(2 * 3 + 2) * NP + p] * t2;
        } else {           // rows (ja,p) of E_l^T: sum over i1 <= ja
          dv[m][0] -= cE[(0 * 3 + 0) * NP + p] * t0;
          dv[m][1] -= cE[(0 * 3 + 1) * NP + p] * t0 + cE[(1 * 3 + 1) * NP + p] * t1;
          dv[m][2] -= cE[(0 * 3 + 2) * NP + p] * t0 + cE[(1 * 3 + 2) * NP + p] * t1 + cE[(2 * 3 + 2) * NP + p] * t2;
          if (jn >= 0) {   // rows (ia,p) of -E_r(ip)^T: sum over j1 >= ia
            // (streaming stores: nobody in this kernel reads them back, and they should not push G out of the L2;
            //  measured: 86 -> 82.7 us.  The same on the W / U / D stores of the other kernels LOSES 2 %: their
            //  consumers are the next launches, which find recently written lines in the caches.)
            __builtin_nontemporal_store(-(cRp[(0 * 3 + 0) * NP + p] * t0 + cRp[(0 * 3 + 1) * NP + p] * t1 + cRp[(0 * 3 + 2) * NP + p] * t2), &Cj[(0 * NP + p) * BS + c]);
            __builtin_nontemporal_store(-(cRp[(1 * 3 + 1) * NP + p] * t1 + cRp[(1 * 3 + 2) * NP + p] * t2), &Cj[(1 * NP + p) * BS + c]);
            __builtin_nontemporal_store(-(cRp[(2 * 3 + 2) * NP + p] * t2), &Cj[(2 * NP + p) * BS + c]);
          }
        }
      }
    }
    if (side == 1 && jn >= 0)
      for (int e = tid; e < 5 * BS; e += 256) Cj[3 * NP * BS + e] = 0.0;    // padding rows of the coupling
    if (tid < 3 * NP) {        // b_j -= E^T z
      const int ja = tid / NP, pa = tid % NP;
      if (side == 0) {
        for (int j1 = ja; j1 < 3; ++j1) bs -= cE[(ja * 3 + j1) * NP + pa] * yv[j1 * NP + pa];
      } else {
        for (int i1 = 0; i1 <= ja; ++i1) bs -= cE[(i1 * 3 + ja) * NP + pa] * yv[i1 * NP + pa];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int q = tid + 256 * m;
    if (q < NP * BS) {
      const int p = q / BS, c = q % BS;
#pragma unroll
      for (int a = 0; a < 3; ++a) __builtin_nontemporal_store(dv[m][a], &Dj[(a * NP + p) * BS + c]);
    }
  }
  if (tid < BS) ch.b[(size_t)j * BS + tid] = bs;
}

// Level-0 back-substitution: x_i = z_i - G_i (E_l x_l + E_r x_r).  G is read straight from HBM into registers - no LDS
// staging, so every workgroup of the level is resident at once and all of its loads are in flight together.  G is
// symmetric: thread (r, part) accumulates over ROWS c of COLUMN r, so the lanes of a wave read consecutive addresses.
__global__ void __launch_bounds__(256)
k_bcr_backsub0(BcrChain ch, const int* __restrict__ elim, const FteConst* __restrict__ cst,
               const int* __restrict__ status) {
  if (status && *status != 0) return;
  __shared__ double xl[BS], xr[BS], vv[BS], ysc[3 * BS], cL[9 * NP], cR[9 * NP];
  const int tid = threadIdx.x;
  const int i = elim[3 * blockIdx.x], l = elim[3 * blockIdx.x + 1], r = elim[3 * blockIdx.x + 2];
  const size_t MB = (size_t)BS * BS;
  const int col = tid % BS, part = tid / BS, c0 = 27 * part, nc = part < 2 ? 27 : 26;
  double g[27];
  if (tid < 3 * BS) {
    const double* G = ch.D + i * MB;
#pragma unroll
    for (int k = 0; k < 27; ++k) g[k] = k < nc ? G[(size_t)(c0 + k) * BS + col] : 0.0;
  }
  fill_coupling_coef(cL, cR, *cst, i, tid);
  if (tid < BS) {
    xl[tid] = l >= 0 ? ch.b[(size_t)l * BS + tid] : 0.0;
    xr[tid] = r >= 0 ? ch.b[(size_t)r * BS + tid] : 0.0;
  }
  const double zi = tid < BS ? ch.b[(size_t)i * BS + tid] : 0.0;
  __syncthreads();
  if (tid < BS) {
    double v = 0.0;
    if (tid < 3 * NP) {
      const int a = tid / NP, p = tid % NP;
      for (int jj = a; jj < 3; ++jj) v += cL[(a * 3 + jj) * NP + p] * xl[jj * NP + p];   // (E_l x_l)[(ii=a,p)]
      for (int ii = 0; ii <= a; ++ii) v += cR[(ii * 3 + a) * NP + p] * xr[ii * NP + p];  // (E_r x_r)[(jj=a,p)]
    }
    vv[tid] = v;
  }
  __syncthreads();
  if (tid < 3 * BS) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 26; k += 2) {
      s0 += g[k] * vv[c0 + k];
      s1 += g[k + 1] * vv[c0 + k + 1];
    }
    if (nc == 27) s0 += g[26] * vv[c0 + 26];
    ysc[tid] = s0 + s1;
  }
  __syncthreads();
  if (tid < BS) ch.b[(size_t)i * BS + tid] = zi - ((ysc[tid] + ysc[BS + tid]) + ysc[2 * BS + tid]);
}

// ---- incomplete reduction: size of the dropped couplings -----------------------------------------------------
// After the isolated last level U[a] = U_a = L_a^-T and U[b] = U_b for the two ends of every dropped coupling block
// C = block(b, a) (stored at Cpl[a]).  Dropping C perturbs the remaining system by E = [[0, C^T], [C, 0]]; in the energy
// norm of the kept block-diagonal part that is || L_b^-1 C L_a^-T ||_2 <= || U_b^T C U_a ||_F =: eps(a, b), and the solve's
// relative error in that norm is <= eps / (1 - eps).  One workgroup per pair: T = C U_a, then N = U_b^T T on the matrix
// cores (U upper triangular: the k loops stop at the diagonal tile), eps^2 = sum N^2 -> trunc_eps2[pair].
__global__ void __launch_bounds__(256)
k_bcr_trunc_check(BcrChain ch, const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Cm = reinterpret_cast<double*>(smem_raw);
  double* Ua = Cm + MAT;
  double* Ub = Ua + MAT;
  double* red = Ub + MAT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int a = ch.d_pairs[2 * blockIdx.x], b = ch.d_pairs[2 * blockIdx.x + 1];
  const size_t MB = (size_t)BS * BS;
  load_mat(Cm, ch.Cpl + a * MB, tid);
  load_mat(Ua, ch.U + a * MB, tid);
  load_mat(Ub, ch.U + b * MB, tid);
  __syncthreads();
  // (the elimination stores the factor's LOWER tiles as workspace: only entries on or above the diagonal are U)
  auto up = [](const double* U, int r, int c) { return c >= r ? U[r * LD + c] : 0.0; };
  // phase 1: T(ib, jb) = sum_{k <= jb} C(ib, k) U_a(k, jb); tiles dealt to the waves, kept in registers
  d4 acc[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int t = wave + 4 * q;
    d4 c4 = {0, 0, 0, 0};
    if (t < 25) {
      const int ib = t / 5, jb = t % 5;
      for (int k4 = 0; k4 < 4 * (jb + 1); ++k4) {
        const int kk = 4 * k4 + lk;
        c4 = mfma(Cm[(ib * 16 + li) * LD + kk], up(Ua, kk, jb * 16 + li), c4);
      }
    }
    acc[q] = c4;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int t = wave + 4 * q;
    if (t < 25) {
      const int ib = t / 5, jb = t % 5;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Cm[(ib * 16 + lk + 4 * rr) * LD + jb * 16 + li] = acc[q][rr];
    }
  }
  __syncthreads();
  // phase 2: N(ib, jb) = sum_{k <= ib} U_b(k, ib)^T T(k, jb)
  double s2 = 0.0;
  for (int t = wave; t < 25; t += 4) {
    const int ib = t / 5, jb = t % 5;
    d4 c4 = {0, 0, 0, 0};
    for (int k4 = 0; k4 < 4 * (ib + 1); ++k4) {
      const int kk = 4 * k4 + lk;
      c4 = mfma(up(Ub, kk, ib * 16 + li), Cm[kk * LD + jb * 16 + li], c4);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) s2 += c4[rr] * c4[rr];
  }
  for (int off = 32; off > 0; off >>= 1) s2 += __shfl_down(s2, off, 64);
  if (lane == 0) red[wave] = s2;
  __syncthreads();
  if (tid == 0) ch.trunc_eps2[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- incomplete reduction: block-Jacobi refinement --------------------------------------------------------------
// The truncated solve x0_j = D_j^-1 b_j ignores the couplings C between the R isolated nodes.  One sweep re-introduces them:
//   x_j <- x0_j - D_j^-1 ( block(j, l) x_l + block(j, r) x_r ),      D_j^-1 = U_j U_j^T,
// with the neighbours' values of the PREVIOUS sweep (Jacobi: every node in parallel, deterministic).  The iteration matrix
// has norm <= 2 eps in the energy norm of the block diagonal (eps = the measured size of the dropped blocks,
// k_bcr_trunc_check), so r sweeps leave a relative error <= (2 eps)^(r+1) / (1 - 2 eps): "exact to rounding" after a few
// sweeps wherever the couplings have decayed, at ~5 us per sweep instead of ~30 us per further reduction level.
// block(j, l) = Cpl[l] (rows j, columns l); block(j, r) = Cpl[j]^T.  Entry p of the isolated level is node iso[3 p].
// src / dst: iterates indexed by p (src == nullptr: the truncated solve's x0 is read from ch.b and saved to x0);
// to_chain: the new iterate also goes to ch.b (only legal when src != nullptr: nobody reads ch.b then).
// (512 threads: the launch is <= ~60 workgroups of three dependent mat-vecs - more waves per workgroup shorten each)
constexpr int RF_T = 512, RF_P = RF_T / BS, RF_W = (BS + RF_P - 1) / RF_P, RF_V = (BS * BS / 2 + RF_T - 1) / RF_T;
__global__ void __launch_bounds__(RF_T)
k_bcr_refine(BcrChain ch, const int* __restrict__ iso, int n_iso, const double* __restrict__ src, double* __restrict__ dst,
             double* __restrict__ x0, int to_chain, double* __restrict__ norms, const int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Ml = reinterpret_cast<double*>(smem_raw);
  double* Mr = Ml + MAT;
  double* Mu = Mr + MAT;
  double* xl = Mu + MAT;          // [80] each
  double* xr = xl + BS;
  double* tv = xr + BS;
  double* part = tv + BS;         // [RF_P][80]
  const int tid = threadIdx.x, p = blockIdx.x;
  const int j = iso[3 * p], l = p > 0 ? iso[3 * (p - 1)] : -1, r = p + 1 < n_iso ? iso[3 * (p + 1)] : -1;
  const size_t MB = (size_t)BS * BS;
  {  // all three matrices in flight before the first LDS write
    const double2* s0 = reinterpret_cast<const double2*>(ch.Cpl + (l >= 0 ? (size_t)l : 0) * MB);
    const double2* s1 = reinterpret_cast<const double2*>(ch.Cpl + (size_t)j * MB);
    const double2* s2 = reinterpret_cast<const double2*>(ch.U + (size_t)j * MB);
    double2 v0[RF_V], v1[RF_V], v2[RF_V];
#pragma unroll
    for (int k = 0; k < RF_V; ++k) {
      const int idx = tid + RF_T * k;
      if (idx < BS * BS / 2) {
        v0[k] = l >= 0 ? s0[idx] : make_double2(0.0, 0.0);
        v1[k] = r >= 0 ? s1[idx] : make_double2(0.0, 0.0);
        v2[k] = s2[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < RF_V; ++k) {
      const int idx = tid + RF_T * k;
      if (idx < BS * BS / 2) {
        const int e = 2 * idx, rr = e / BS, c = e % BS;
        Ml[rr * LD + c] = v0[k].x;  Ml[rr * LD + c + 1] = v0[k].y;
        Mr[rr * LD + c] = v1[k].x;  Mr[rr * LD + c + 1] = v1[k].y;
        Mu[rr * LD + c] = v2[k].x;  Mu[rr * LD + c + 1] = v2[k].y;
      }
    }
  }
  double xj0 = 0.0;
  if (tid < BS) {
    if (src) {
      xl[tid] = l >= 0 ? src[(size_t)(p - 1) * BS + tid] : 0.0;
      xr[tid] = r >= 0 ? src[(size_t)(p + 1) * BS + tid] : 0.0;
      xj0 = x0[(size_t)p * BS + tid];
    } else {
      xl[tid] = l >= 0 ? ch.b[(size_t)l * BS + tid] : 0.0;
      xr[tid] = r >= 0 ? ch.b[(size_t)r * BS + tid] : 0.0;
      xj0 = ch.b[(size_t)j * BS + tid];
      x0[(size_t)p * BS + tid] = xj0;
    }
  }
  __syncthreads();
  const int row = tid % BS, pr = tid / BS, k0 = RF_W * pr, k1 = min(k0 + RF_W, BS);
  auto row_sum = [&](int r) {     // the RF_P partial sums of row r, fixed order
    double v = part[r];
#pragma unroll
    for (int q = 1; q < RF_P; ++q) v += part[q * BS + r];
    return v;
  };
  if (tid < RF_P * BS) {             // t = block(j, l) x_l + block(j, r) x_r  (fixed trip count: the LDS reads pipeline)
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < RF_W; ++q) {
      const int k = k0 + (k0 + q < k1 ? q : 0);
      const double w = k0 + q < k1 ? 1.0 : 0.0;
      s0 += w * Ml[row * LD + k] * xl[k];
      s1 += w * Mr[k * LD + row] * xr[k];
    }
    part[tid] = s0 + s1;
  }
  __syncthreads();
  if (tid < BS) tv[tid] = row_sum(tid);
  __syncthreads();
  if (tid < RF_P * BS) {             // w = U^T t  (U upper triangular: rows <= column)
    double s = 0.0;
    const int c1 = min(k1, row + 1);
#pragma unroll
    for (int q = 0; q < RF_W; ++q) {
      const bool on = k0 + q < c1;
      const int k = k0 + (on ? q : 0);
      s += (on ? 1.0 : 0.0) * Mu[k * LD + row] * tv[k];
    }
    part[tid] = s;
  }
  __syncthreads();
  if (tid < BS) xl[tid] = row_sum(tid);      // (xl reused: w)
  __syncthreads();
  if (tid < RF_P * BS) {             // d = U w
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < RF_W; ++q) {
      const bool on = k0 + q >= row && k0 + q < k1;
      const int k = on ? k0 + q : row;
      s += (on ? 1.0 : 0.0) * Mu[row * LD + k] * xl[k];
    }
    part[tid] = s;
  }
  __syncthreads();
  double dabs = 0.0, xabs = 0.0;
  if (tid < BS) {
    const double x = xj0 - row_sum(tid);
    dst[(size_t)p * BS + tid] = x;
    if (to_chain) ch.b[(size_t)j * BS + tid] = x;
    // size of this sweep's update of the node (against its own previous value) and of the solution
    const double xprev = src ? src[(size_t)p * BS + tid] : xj0;
    dabs = fabs(x - xprev);
    xabs = fabs(x);
  }
  if (norms) {                    // (waves 0 and 1 hold the 80 rows)
    for (int off = 32; off > 0; off >>= 1) {
      dabs = fmax(dabs, __shfl_down(dabs, off, 64));
      xabs = fmax(xabs, __shfl_down(xabs, off, 64));
    }
    __syncthreads();
    if ((tid & 63) == 0 && tid < 128) {
      part[2 * (tid >> 6)] = dabs;
      part[2 * (tid >> 6) + 1] = xabs;
    }
    __syncthreads();
    if (tid == 0) {
      norms[p] = fmax(part[0], part[2]);
      norms[n_iso + p] = fmax(part[1], part[3]);
    }
  }
}
__global__ void k_bcr_refine_copy(BcrChain ch, const int* __restrict__ iso, const double* __restrict__ src,
                                  const int* __restrict__ status) {
  if (status && *status != 0) return;
  if (threadIdx.x < BS) ch.b[(size_t)iso[3 * blockIdx.x] * BS + threadIdx.x] = src[(size_t)blockIdx.x * BS + threadIdx.x];
}

// ---- the whole back-substitution of a truncated + refined reduction as ONE launch -------------------------------------
// (separator chain of the chunked solver: <= ~240 nodes, every kernel of it a latency chain.)  What were 2 + r + K launches
// - truncated solve of the isolated nodes, r block-Jacobi sweeps, K back-substitution levels, each re-reading its three
// 51 KB matrices - is one workgroup per node that loads its matrices ONCE into LDS and then only trades 80-double vectors
// with its neighbours through memory:
//   isolated node p (blocks 0 .. n_iso-1):  x0 = U y;  sweep s = 1..r:  x(s) = x0 - U U^T (C_l x_l(s-1) + C_r x_r(s-1));
//     version s of the node's iterate goes to buffer s & 1 of the node, every value in a pair of 64-bit words that carry the
//     version's tag next to the data (ll_put / ll_get below: no flag, no fence); a neighbour writes version s only after it
//     has read this node's version s - 1, so two buffers suffice.
//   node i of level k < K (blocks after them, deepest level first):  x_i = U (y_i - W_l x_l - W_r x_r) once done[l], done[r].
// Isolated workgroups wait for EACH OTHER, so all n_iso of them must be resident together: one per CU (156 KB of LDS), the
// lowest block indices of the launch - the host only uses this kernel for n_iso <= 128 on a GPU that is not shared
// (acino_fte_params::shared_gpu = 0); every other workgroup waits for lower block indices only.  Waits are bounded polls
// (as in k_bcr_backsub_tail): a timeout flags the step (numeric_err bit 1 -> status 6) and lets the launch drain.
// The one flag left (ver[0], below) is zeroed and the epoch of the tags is advanced by the consumer that follows
// (k_chunk_backsub), i.e. before the next launch of this kernel.
constexpr int ST_T = 512, ST_P = ST_T / BS, ST_W = (BS + ST_P - 1) / ST_P, ST_V = (BS * BS / 2 + ST_T - 1) / ST_T;
struct SepTailArgs {
  const int* iso;        // [n_iso][3] entries of the isolated level (node, -1, -1)
  int n_iso, refine;
  int n_lv;              // regular levels below the isolated one
  int lv_off[12];        // elim-entry offset of level K-1, K-2, ... 0
  int lv_cnt[12];
  int* ver;              // ver[0]: isolated node 0 has published its truncated solve (the level nodes hold their loads back until then)
  const int* epoch;      // the launch's number: part of every tag, so the slots never need cleaning
  unsigned long long* ll;   // [2][n_iso][80][2] iterates of the sweeps, then [n_nodes][80][2] solutions - tagged pairs
  double* norms;         // [4][n_iso]: |update|, |x| of the last sweep; the same of the sweep before it
  // chains with fused narrow levels: the isolated nodes are FACTORED HERE (no elimination launch for them, no round trip of
  // the factor): D + AL + SL + SR is formed in LDS; iso_loc = per entry (flags, location of block(next isolated node, this one))
  const int* iso_loc;
  int fused;
};
// Hand-off of an 80-double vector between workgroups of one launch.  Every double travels as two 64-bit words, each half of
// the value next to a 32-bit tag (epoch of the launch, version of the vector), written and read as agent-scope RELAXED atomics
// (written through to / read from the point where the XCDs' L2s are coherent).  A 64-bit access is single-copy atomic, so a
// word's tag vouches for the half beside it: the reader polls the words themselves until both carry the tag it expects.  No
// flag behind the data, hence nothing that has to be ordered: no wait for the stores' acknowledgement, no barrier, no second
// round trip to read the values after the flag (the flag protocol this replaces: ~1.6 us to publish - vmcnt(0), barrier, flag -
// and ~0.6 us between seeing the flag and having the values, per hand-off, by the stamps of scripts/sep_stamps.py; and no
// release / acquire fences, which write back / invalidate a whole L2: ~1.3 us + ~0.5 us each).  Tags are compared for equality
// and differ between consecutive launches and between the versions of one launch, so a stale slot can never pass and the slots
// are never cleaned.
__device__ __forceinline__ void ll_put(unsigned long long* slot, double v, unsigned tag) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
  __hip_atomic_store(slot, (u & 0xffffffffull) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(slot + 1, (u >> 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ll_get(const unsigned long long* slot, unsigned tag, int* numeric_err) {
  long long polls = 0;
  for (;;) {
    const unsigned long long w0 = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag)
      return __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
    __builtin_amdgcn_s_sleep(1);
    if (++polls > (1ll << 22)) {      // (bit 3: it was THIS kernel - the host may fall back to the per-level kernels)
      if (numeric_err) atomicOr(numeric_err, 2 | 8);
      return 0.0;
    }
  }
}
// both neighbours' vectors into LDS: waves 0-1 poll the left one's slots, waves 2-3 the right one's (null: zeros)
__device__ __forceinline__ void ll_get_pair(double* xl, double* xr, const unsigned long long* sl, const unsigned long long* sr,
                                            unsigned tag, int* numeric_err, int tid) {
  if (tid < BS) xl[tid] = sl ? ll_get(sl + 2 * tid, tag, numeric_err) : 0.0;
  else if (tid >= 128 && tid < 128 + BS) xr[tid - 128] = sr ? ll_get(sr + 2 * (tid - 128), tag, numeric_err) : 0.0;
}
__device__ __forceinline__ void st_raise(int* flag, int v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wait(const int* flag, int need, int* numeric_err) {
  long long polls = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++polls > (1ll << 22)) {
      if (numeric_err) atomicOr(numeric_err, 2 | 8);      // (bit 3: it was THIS kernel - the host may fall back to the per-level kernels)
      break;
    }
  }
}
__global__ void __launch_bounds__(ST_T)
k_sep_tail(BcrChain ch, SepTailArgs a, const int* __restrict__ status, int* __restrict__ numeric_err) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (status && *status != 0) return;
  double* Ml = reinterpret_cast<double*>(smem_raw);
  double* Mr = Ml + MAT;
  double* Mu = Mr + MAT;
  double* xl = Mu + MAT;          // [80] each
  double* xr = xl + BS;
  double* tv = xr + BS;
  double* part = tv + BS;         // [ST_P][80]
  const int tid = threadIdx.x;
  const size_t MB = (size_t)BS * BS;
  const unsigned tag0 = (unsigned)*a.epoch << 6;          // tags of this launch: + 1 + s iterate of sweep s, + 63 a node's solution
  unsigned long long* const fin = a.ll + (size_t)2 * a.n_iso * BS * 2;
#define ST_STAMP(k) do { if (ch.dbg && tid == 0 && (long long)blockIdx.x == ch.dbg[64] && ch.dbg[65] == 100) ch.dbg[k] = (long long)wall_clock64(); } while (0)
  ST_STAMP(6);
  if (ch.dbg && tid == 0 && blockIdx.x == 0 && ch.dbg[65] == 100) ch.dbg[7] = (long long)wall_clock64();    // (workgroup 0's start: the common time base)
  const int row = tid % BS, pr = tid / BS, k0 = ST_W * pr, k1 = min(k0 + ST_W, BS);
  auto row_sum = [&](int r) {     // the ST_P partial sums of row r, fixed order
    double v = part[r];
#pragma unroll
    for (int q = 1; q < ST_P; ++q) v += part[q * BS + r];
    return v;
  };
  auto load3 = [&](const double* m0, const double* m1, const double* m2) {   // all three matrices in flight before the first LDS write
    const double2* s0 = reinterpret_cast<const double2*>(m0);
    const double2* s1 = reinterpret_cast<const double2*>(m1);
    const double2* s2 = reinterpret_cast<const double2*>(m2);
    double2 v0[ST_V], v1[ST_V], v2[ST_V];
#pragma unroll
    for (int k = 0; k < ST_V; ++k) {
      const int idx = tid + ST_T * k;
      if (idx < BS * BS / 2) {
        v0[k] = m0 ? s0[idx] : make_double2(0.0, 0.0);
        v1[k] = m1 ? s1[idx] : make_double2(0.0, 0.0);
        v2[k] = s2[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < ST_V; ++k) {
      const int idx = tid + ST_T * k;
      if (idx < BS * BS / 2) {
        const int e = 2 * idx, rr = e / BS, c = e % BS;
        Ml[rr * LD + c] = v0[k].x;  Ml[rr * LD + c + 1] = v0[k].y;
        Mr[rr * LD + c] = v1[k].x;  Mr[rr * LD + c + 1] = v1[k].y;
        Mu[rr * LD + c] = v2[k].x;  Mu[rr * LD + c + 1] = v2[k].y;
      }
    }
  };
  // x = U t (U upper triangular, t in tv) -> value of row tid for tid < 80 (two barriers inside)
  auto upper_matvec = [&]() {
    if (tid < ST_P * BS) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < ST_W; ++q) {
        const bool on = k0 + q >= row && k0 + q < k1;
        const int k = on ? k0 + q : row;
        s += (on ? 1.0 : 0.0) * Mu[row * LD + k] * tv[k];
      }
      part[tid] = s;
    }
    __syncthreads();
    const double v = tid < BS ? row_sum(tid) : 0.0;
    __syncthreads();
    return v;
  };
  // x = G t for a FULL symmetric G in Mu (fused isolated level: G = D^-1 replaces the factor, one product instead of two)
  auto full_matvec = [&]() {
    if (tid < ST_P * BS) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < ST_W; ++q) {
        const int k = k0 + (k0 + q < k1 ? q : 0);
        s += (k0 + q < k1 ? 1.0 : 0.0) * Mu[row * LD + k] * tv[k];
      }
      part[tid] = s;
    }
    __syncthreads();
    const double v = tid < BS ? row_sum(tid) : 0.0;
    __syncthreads();
    return v;
  };
  if ((int)blockIdx.x < a.n_iso) {
    // ---------------- isolated node: truncated solve + refinement sweeps ----------------
    const int p = blockIdx.x, n_iso = a.n_iso;
    const int j = a.iso[3 * p], l = p > 0 ? a.iso[3 * (p - 1)] : -1, r = p + 1 < n_iso ? a.iso[3 * (p + 1)] : -1;
    ST_STAMP(0);
    if (a.fused) {
      const int fl = a.iso_loc[2 * p];
      const double2* s0 = reinterpret_cast<const double2*>(ch.Cpl + (size_t)(l >= 0 ? a.iso_loc[2 * (p - 1) + 1] : 0) * MB);
      const double2* s1 = reinterpret_cast<const double2*>(ch.Cpl + (size_t)(r >= 0 ? a.iso_loc[2 * p + 1] : 0) * MB);
      double2 v0[ST_V], v1[ST_V];
#pragma unroll
      for (int k = 0; k < ST_V; ++k) {
        const int idx = tid + ST_T * k;
        if (idx < BS * BS / 2) {
          v0[k] = l >= 0 ? s0[idx] : make_double2(0.0, 0.0);
          v1[k] = r >= 0 ? s1[idx] : make_double2(0.0, 0.0);
        }
      }
      load_node_sum<ST_T>(Mu, tv, ch, j, fl, tid);               // D_j with everything the levels below left for it; b_j -> tv
#pragma unroll
      for (int k = 0; k < ST_V; ++k) {
        const int idx = tid + ST_T * k;
        if (idx < BS * BS / 2) {
          const int e = 2 * idx, rr = e / BS, c = e % BS;
          Ml[rr * LD + c] = v0[k].x;  Ml[rr * LD + c + 1] = v0[k].y;
          Mr[rr * LD + c] = v1[k].x;  Mr[rr * LD + c + 1] = v1[k].y;
        }
      }
      __syncthreads();
      ST_STAMP(1);
      chol80<ST_T / 64>(Mu, tid, numeric_err);
      ST_STAMP(2);
      // G = D^-1 = U U^T over the factor (nobody else needs the factor of an isolated node): every later solve with D is ONE
      // 80 x 80 product instead of two triangular ones - the truncated solve and each of the sweeps lose a dependent stage
      // (adding each column block's term under the factorisation, through chol80's hook, was tried: the factorisation grows by
      //  what the product shrinks, 0.9 us each - the terms are few and the hooks' LDS traffic sits beside the pivot chain's)
      {
        const int wv = tid >> 6, ln = tid & 63, li = ln & 15, lk = ln >> 4;
        d4 gacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = wv + 8 * u;
          gacc[u] = d4{0, 0, 0, 0};
          if (t < 15) {
            const int ib = tri_i(t), jb = tri_j(t);
            const double* pa = Mu + (ib * 16 + li) * LD + ib * 16 + lk;   // U(ib, k >= ib)[i][kk]
            const double* pb = Mu + (jb * 16 + li) * LD + ib * 16 + lk;   // U(jb, k >= ib)[j][kk]
            switch (ib) {
              case 0: gacc[u] = mma_seq<20, false>(gacc[u], pa, 4, pb, 4); break;
              case 1: gacc[u] = mma_seq<16, false>(gacc[u], pa, 4, pb, 4); break;
              case 2: gacc[u] = mma_seq<12, false>(gacc[u], pa, 4, pb, 4); break;
              case 3: gacc[u] = mma_seq<8, false>(gacc[u], pa, 4, pb, 4); break;
              default: gacc[u] = mma_seq<4, false>(gacc[u], pa, 4, pb, 4); break;
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = wv + 8 * u;
          if (t < 15) {
            const int ib = tri_i(t), jb = tri_j(t);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              Mu[(ib * 16 + lk + 4 * rr) * LD + jb * 16 + li] = gacc[u][rr];
              if (ib != jb) Mu[(jb * 16 + li) * LD + ib * 16 + lk + 4 * rr] = gacc[u][rr];
            }
          }
        }
      }
    } else {
      load3(l >= 0 ? ch.Cpl + (size_t)l * MB : nullptr, r >= 0 ? ch.Cpl + (size_t)j * MB : nullptr, ch.U + (size_t)j * MB);
      if (tid < BS) tv[tid] = ybuf(ch)[(size_t)j * BS + tid];     // y_j = U^T b_j (written by the reduction)
    }
    __syncthreads();
    const bool use_g = a.fused != 0;
    const double x0 = use_g ? full_matvec() : upper_matvec();    // truncated solve (fused: G b; else U y)
    ST_STAMP(3);
    double xcur = x0, dabs = 0.0, dabs_prev = 0.0, xabs = fabs(x0), xabs_prev = 0.0, d_first = 0.0, d_second = 0.0;
    unsigned long long* xb0 = a.ll;
    unsigned long long* xb1 = a.ll + (size_t)n_iso * BS * 2;
    if (a.refine > 0) {
      if (tid < BS) ll_put(xb0 + ((size_t)p * BS + tid) * 2, x0, tag0 + 1);
      if (p == 0 && tid == 0) st_raise(a.ver, 1);          // (a throttle, not a hand-off: nothing is read behind it)
    }
    for (int s = 1; s <= a.refine; ++s) {
      const unsigned long long* src = (s - 1) & 1 ? xb1 : xb0;
      unsigned long long* dst = s & 1 ? xb1 : xb0;
      if (s == 2) ST_STAMP(8);
      ll_get_pair(xl, xr, l >= 0 ? src + (size_t)(p - 1) * BS * 2 : nullptr, r >= 0 ? src + (size_t)(p + 1) * BS * 2 : nullptr,
                  tag0 + (unsigned)s, numeric_err, tid);
      __syncthreads();
      if (s == 2) ST_STAMP(11);
      if (tid < ST_P * BS) {         // t = block(j, l) x_l + block(j, r) x_r ;  block(j, r) = Cpl[j]^T
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < ST_W; ++q) {
          const int k = k0 + (k0 + q < k1 ? q : 0);
          const double w = k0 + q < k1 ? 1.0 : 0.0;
          s0 += w * Ml[row * LD + k] * xl[k];
          s1 += w * Mr[k * LD + row] * xr[k];
        }
        part[tid] = s0 + s1;
      }
      __syncthreads();
      if (tid < BS) tv[tid] = row_sum(tid);
      __syncthreads();
      double d;
      if (use_g) {
        d = full_matvec();                                       // d = G t
      } else {
        if (tid < ST_P * BS) {       // w = U^T t  (rows <= column)
          double sm = 0.0;
          const int c1 = min(k1, row + 1);
#pragma unroll
          for (int q = 0; q < ST_W; ++q) {
            const bool on = k0 + q < c1;
            const int k = k0 + (on ? q : 0);
            sm += (on ? 1.0 : 0.0) * Mu[k * LD + row] * tv[k];
          }
          part[tid] = sm;
        }
        __syncthreads();
        const double w = tid < BS ? row_sum(tid) : 0.0;
        __syncthreads();
        if (tid < BS) tv[tid] = w;
        __syncthreads();
        d = upper_matvec();                                      // d = U w
      }
      if (tid < BS) {
        const double x = x0 - d;
        dabs_prev = dabs;
        xabs_prev = xabs;
        dabs = fabs(x - xcur);
        xabs = fabs(x);
        xcur = x;
        if (s == 1) d_first = dabs;
        if (s == 2) d_second = dabs;
        if (s < a.refine) ll_put(dst + ((size_t)p * BS + tid) * 2, x, tag0 + 1 + (unsigned)s);
      }
      if (s == 2) ST_STAMP(12);
    }
    if (tid < BS) {
      ll_put(fin + ((size_t)j * BS + tid) * 2, xcur, tag0 + 63);
      ch.b[(size_t)j * BS + tid] = xcur;
    }
    ST_STAMP(4);
    if (a.norms && a.refine > 0) {            // (waves 0 and 1 hold the 80 rows)
      for (int off = 32; off > 0; off >>= 1) {
        dabs = fmax(dabs, __shfl_down(dabs, off, 64));
        xabs = fmax(xabs, __shfl_down(xabs, off, 64));
        dabs_prev = fmax(dabs_prev, __shfl_down(dabs_prev, off, 64));
        xabs_prev = fmax(xabs_prev, __shfl_down(xabs_prev, off, 64));
        d_first = fmax(d_first, __shfl_down(d_first, off, 64));
        d_second = fmax(d_second, __shfl_down(d_second, off, 64));
      }
      if ((tid & 63) == 0 && tid < 128) {
        double* q6 = part + 6 * (tid >> 6);
        q6[0] = dabs;
        q6[1] = xabs;
        q6[2] = dabs_prev;
        q6[3] = xabs_prev;
        q6[4] = d_first;
        q6[5] = d_second;
      }
      __syncthreads();
      if (tid == 0) {             // (layout: bcr_backsub, the per-sweep launches)
        a.norms[p] = fmax(part[0], part[6]);
        a.norms[n_iso + p] = fmax(part[1], part[7]);
        if (a.refine >= 2) {
          a.norms[2 * n_iso + p] = fmax(part[2], part[8]);
          a.norms[3 * n_iso + p] = fmax(part[3], part[9]);
        }
        if (a.refine >= 3) a.norms[4 * n_iso + p] = fmax(part[4], part[10]);
        if (a.refine >= 4) a.norms[5 * n_iso + p] = fmax(part[5], part[11]);
      }
    }
    return;
  }
  // ---------------- node of a regular level ----------------
  int q = (int)blockIdx.x - a.n_iso, lvl = 0;
  while (lvl < a.n_lv && q >= a.lv_cnt[lvl]) q -= a.lv_cnt[lvl++];
  if (lvl >= a.n_lv) return;
  const int* en = ch.d_elim + 3 * (a.lv_off[lvl] + q);
  const int i = en[0], l = en[1], r = en[2];
  if (a.fused && a.refine > 0) {
    // The isolated workgroups of this launch are the critical path and start with ~200 KB of loads each; the 3 x 51 KB of every
    // level node are not needed before the sweeps are over.  Hold them back until the first isolated node has published its
    // truncated solve (its factorisation is done by then), so the two sets of loads do not share the memory system.
    if (tid == 0) st_wait(a.ver, 1, nullptr);
    __syncthreads();
  }
  load3(l >= 0 ? ch.Wl + (size_t)i * MB : nullptr, r >= 0 ? ch.Wr + (size_t)i * MB : nullptr, ch.U + (size_t)i * MB);
  const double yi = tid < BS ? ybuf(ch)[(size_t)i * BS + tid] : 0.0;
  ll_get_pair(xl, xr, l >= 0 ? fin + (size_t)l * BS * 2 : nullptr, r >= 0 ? fin + (size_t)r * BS * 2 : nullptr, tag0 + 63,
              numeric_err, tid);
  __syncthreads();
  ST_STAMP(1);                  // (level node: both neighbours' solutions in LDS)
  if (tid < ST_P * BS) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int qq = 0; qq < ST_W; ++qq) {
      const int k = k0 + (k0 + qq < k1 ? qq : 0);
      const double w = k0 + qq < k1 ? 1.0 : 0.0;
      s0 += w * Ml[row * LD + k] * xl[k];
      s1 += w * Mr[row * LD + k] * xr[k];
    }
    part[tid] = s0 + s1;
  }
  __syncthreads();
  const double t = tid < BS ? yi - row_sum(tid) : 0.0;
  __syncthreads();
  if (tid < BS) tv[tid] = t;
  __syncthreads();
  const double x = upper_matvec();
  if (tid < BS) {
    ll_put(fin + ((size_t)i * BS + tid) * 2, x, tag0 + 63);
    ch.b[(size_t)i * BS + tid] = x;
  }
  ST_STAMP(5);
#undef ST_STAMP
}

// ---- host side ------------------------------------------------------------------------------
void BcrSchedule::build(int n, bool pin_left, bool pin_right, int max_levels, int refine_sweeps, bool fused) {
  refine = 0;
  levels.clear();
  elim.clear();
  remain.clear();
  pairs.clear();
  elim6.clear();
  iso_loc.clear();
  fold.clear();
  n_fold_pins = 0;
  fused_levels = fused;
  // fused narrow levels: which running sums of a node hold something, and where the block that couples a live node to the
  // next live node on its right is (slot `node` of the coupling array, or n + i once the elimination of i has created it)
  std::vector<char> has_sl(n, 0), has_sr(n, 0);
  std::vector<int> rloc(n);
  for (int i = 0; i < n; ++i) rloc[i] = i;
  std::vector<int> fold_iso;
  std::vector<int> act(n);
  for (int i = 0; i < n; ++i) act[i] = i;
  auto pinned = [&](int node) { return (pin_left && node == 0) || (pin_right && node == n - 1); };
  while (true) {
    const int R = (int)act.size();
    if (max_levels > 0 && (int)levels.size() == max_levels && R > 1 && !pin_left && !pin_right) {
      // incomplete reduction: drop the couplings between the R remaining nodes, solve each on its own
      BcrLevel lv;
      lv.elim_off = (int)elim.size() / 3;
      lv.remain_off = (int)remain.size() / 4;
      lv.n_elim = R;
      lv.n_remain = 0;
      lv.adjacent = false;
      lv.isolated = true;
      for (int p = 0; p < R; ++p) {
        elim.push_back(act[p]);
        elim.push_back(-1);
        elim.push_back(-1);
        if (p + 1 < R) {
          pairs.push_back(act[p]);
          pairs.push_back(act[p + 1]);
        }
        if (fused) {
          const int fl = (has_sl[act[p]] ? 1 : 0) | (has_sr[act[p]] ? 2 : 0), loc = p + 1 < R ? rloc[act[p]] : -1;
          iso_loc.push_back(fl);
          iso_loc.push_back(loc);
          fold_iso.push_back(act[p]);
          fold_iso.push_back(fl);
          fold_iso.push_back(loc);
        }
      }
      levels.push_back(lv);
      act.clear();
      break;
    }
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
    lv.adjacent = false;
    std::vector<int> next;
    for (int p = 0; p < R; ++p) {
      if (pick[p]) {
        if ((p > 0 && act[p - 1] == act[p] - 1) || (p + 1 < R && act[p + 1] == act[p] + 1)) lv.adjacent = true;
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
      if (im < 0 && ip < 0 && !levels.empty()) continue;   // level 0 lists every remaining node (fused build)
      remain.push_back(act[p]);
      remain.push_back(im);
      remain.push_back(ip);
      remain.push_back(jn);
      ++n_rem;
    }
    lv.n_remain = n_rem;
    if (fused && n_pick <= 128) {          // narrow level: one launch, elimination and Schur products (seplevel.hip)
      lv.fused = true;
      lv.T = std::max(2, slv_workgroups_per_node(n_pick));      // (>= 2: a workgroup's tile share must fit its registers)
      lv.e6_off = (int)elim6.size() / 6;
      for (int p = 0; p < R; ++p) {
        if (!pick[p]) continue;
        const int i = act[p], l = p > 0 ? act[p - 1] : -1, r = p + 1 < R ? act[p + 1] : -1;
        elim6.push_back(i);
        elim6.push_back(l);
        elim6.push_back(r);
        elim6.push_back((has_sl[i] ? 1 : 0) | (has_sr[i] ? 2 : 0) | ((l >= 0 && has_sr[l]) ? 4 : 0) | ((r >= 0 && has_sl[r]) ? 8 : 0));
        elim6.push_back(l >= 0 ? rloc[l] : 0);
        elim6.push_back(r >= 0 ? rloc[i] : 0);
      }
      for (int p = 0; p < R; ++p) {
        if (!pick[p]) continue;
        const int i = act[p], l = p > 0 ? act[p - 1] : -1, r = p + 1 < R ? act[p + 1] : -1;
        if (l >= 0) has_sr[l] = 1;
        if (r >= 0) has_sl[r] = 1;
        if (l >= 0) rloc[l] = r >= 0 ? n + i : -1;
      }
    }
    levels.push_back(lv);
    act.swap(next);
    if (act.empty()) break;
  }
  if (fused) {             // what is left are the pins: their sums are materialised for the export (k_sep_fold)
    for (size_t p = 0; p < act.size(); ++p) {
      fold.push_back(act[p]);
      fold.push_back((has_sl[act[p]] ? 1 : 0) | (has_sr[act[p]] ? 2 : 0));
      fold.push_back(p + 1 < act.size() ? rloc[act[p]] : -1);
    }
    n_fold_pins = (int)act.size();
    fold.insert(fold.end(), fold_iso.begin(), fold_iso.end());
  }
  tail.clear();
  tail_levels = 0;
  int total = 0;
  int k_top = (int)levels.size() - 1;
  if (refine_sweeps > 0 && !levels.empty() && levels.back().isolated) {
    refine = refine_sweeps;                                // the isolated level is refined before the levels above it
    --k_top;                                               // use it: it cannot be part of the fused tail
  }
  for (int k = k_top; k >= 1; --k) {                       // (level 0 of an FTE chain has its own kernel)
    const BcrLevel& lv = levels[k];
    if (total + lv.n_elim > 128) break;
    for (int e = 0; e < lv.n_elim; ++e) {
      const int* en = &elim[3 * (lv.elim_off + e)];
      tail.push_back(en[0]);
      tail.push_back(en[1]);
      tail.push_back(en[2]);
      tail.push_back(total);                              // every entry of the deeper levels comes first
    }
    total += lv.n_elim;
    ++tail_levels;
  }
  if (tail_levels < 2) {                                   // nothing to fuse
    tail.clear();
    tail_levels = 0;
  }
}

static constexpr size_t kElimLds = (MAT + BS + 8 + 18 * NP + 3 * BS) * sizeof(double);
static constexpr size_t kElimDeepLds = (MAT + BS + 3 * BS) * sizeof(double);
static constexpr size_t kUpdateLds = (MAT + BS + 8) * sizeof(double);
static constexpr size_t kUpdateDeepLds = (2 * MAT + 2 * BS + 3 * BS) * sizeof(double);
static constexpr size_t kBacksubLds = (MAT + 9 * BS) * sizeof(double);
static constexpr size_t kBacksubTailLds = (3 * MAT + 9 * BS) * sizeof(double);
static constexpr size_t kUpdate0Lds = (MAT + BS + 8 + 36 * NP) * sizeof(double);
static constexpr size_t kTruncCheckLds = (3 * MAT + 8) * sizeof(double);

int bcr_set_func_attributes() {
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_elim),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kElimLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_elim_deep),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kElimDeepLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_update),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpdateLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_update_deep),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpdateDeepLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_backsub_tail),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubTailLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_backsub<512>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_backsub<256>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_update0),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpdate0Lds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_trunc_check),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTruncCheckLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_refine),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubTailLds));
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sep_tail),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBacksubTailLds));
  return slv_set_func_attributes();
}

// k_sep_tail's isolated workgroups spin-wait on each other's flags, the levels above them on lower block indices: the launch
// is only safe when ALL of its workgroups are resident at once.  blocks = workgroups the schedule would launch (0: the
// kernel does not apply), capacity = what the current device holds of them (occupancy query x compute units - a CU-masked
// or partitioned device, CPX mode, answers with what it really has).
int bcr_sep_tail_fit(const BcrSchedule& sch, int* blocks, int* capacity) {
  *blocks = 0;
  *capacity = 0;
  const int top = (int)sch.levels.size() - 1;
  if (!(sch.refine > 0 && top >= 0 && sch.levels[top].isolated && top <= 12 && sch.levels[top].n_elim <= 128)) return ACINO_OK;
  int n = 0;
  for (int k = top; k >= 0; --k) n += sch.levels[k].n_elim;
  *blocks = n;
  int dev = 0, cus = 0, per_cu = 0;
  ACINO_HIP_CHECK(hipGetDevice(&dev));
  ACINO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  ACINO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_sep_tail), ST_T,
                                                               kBacksubTailLds));
  *capacity = cus * per_cu;
  return ACINO_OK;
}

// Debug (ACINO_DEBUG_SYNC=1): after every launch of the reduction, synchronise and report the first kernel after which
// the numeric-error flag is set or the level's outputs hold a NaN.
static bool dbg_check(const char* what, int level, const BcrChain& ch, const int* ent, int n_ent, int stride, int* d_err,
                      hipStream_t s) {
  if (hipStreamSynchronize(s) != hipSuccess) return true;
  int e = 0;
  (void)hipMemcpy(&e, d_err, sizeof(int), hipMemcpyDeviceToHost);
  std::vector<int> h(n_ent * stride);
  (void)hipMemcpy(h.data(), ent, sizeof(int) * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> m((size_t)BS * BS), bb(BS);
  long long nan_d = 0, nan_b = 0;
  int first = -1;
  for (int k = 0; k < n_ent; ++k) {
    const int node = h[(size_t)k * stride];
    (void)hipMemcpy(m.data(), ch.D + (size_t)node * BS * BS, sizeof(double) * m.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(bb.data(), ch.b + (size_t)node * BS, sizeof(double) * BS, hipMemcpyDeviceToHost);
    long long c = 0;
    for (double v : m) c += (v != v);
    long long cb = 0;
    for (double v : bb) cb += (v != v);
    if ((c || cb) && first < 0) first = node;
    nan_d += c;
    nan_b += cb;
  }
  fprintf(stderr, "[acino debug] level %d %s: numeric_err %d, NaN in D %lld, in b %lld (first node %d of %d)\n", level, what, e,
          nan_d, nan_b, first, n_ent);
  return e != 0 || nan_d || nan_b;
}

bool bcr_level0_adds_al(const BcrSchedule& sch) {
  if (sch.levels.empty() || getenv("ACINO_SEP_COMBINE")) return false;
  if (sch.fused_levels) {      // every node is consumed by a kernel that forms D + AL + SL + SR itself
    for (const BcrLevel& lv : sch.levels)
      if (!lv.fused && !lv.isolated) return false;
    return true;
  }
  const BcrLevel& lv = sch.levels[0];
  // (every node of the chain must pass through exactly one of the two narrow-level kernels at level 0)
  return !lv.isolated && lv.n_elim <= 128 && lv.n_remain <= 128;
}

// true when k_sep_tail runs the back-substitution of this chain (truncated solve, sweeps and every level above in one launch)
static bool sep_tail_applies(const BcrChain& ch, const BcrSchedule& sch) {
  const int top = (int)sch.levels.size() - 1;
  return sch.refine > 0 && sch.refine <= 60 && top >= 0 && sch.levels[top].isolated && ch.refine_buf && ch.st_flags && ch.st_ll && top <= 12 &&
         sch.levels[top].n_elim <= 128;
}

int bcr_reduce(const BcrChain& ch_in, const BcrSchedule& sch, const FteConst* d_c, int* d_numeric_err,
               const int* d_status, hipStream_t s, Profiler* prof) {
  static const bool dbg = getenv("ACINO_DEBUG_SYNC") != nullptr;
  bool dbg_hit = false;
  int level = 0;
  const bool fz = sch.fused_levels && ch_in.SL != nullptr;      // chain with fused narrow levels
  BcrChain ch = ch_in;
  const bool add_al = !fz && ch.AL0 != nullptr && bcr_level0_adds_al(sch);
  for (const BcrLevel& lv : sch.levels) {
    if (fz && lv.fused) {
      {
        ProfSpan sp(prof, PC_ELIM_DEEP, s, lv.n_elim);
        if (int rc = slv_launch_level(ch, lv, d_numeric_err, d_status, s)) return rc;
      }
      if (dbg && !dbg_hit) dbg_hit = dbg_check("level (fused)", level, ch, ch.d_elim + 3 * lv.elim_off, lv.n_elim, 3, d_numeric_err, s);
      ++level;
      continue;
    }
    if (fz && lv.isolated) {
      if (sep_tail_applies(ch, sch)) break;              // k_sep_tail factors the isolated nodes itself
      // per-level kernels: materialise D + AL + SL + SR (and the couplings' home slots) for them
      {
        ProfSpan sp(prof, PC_SEP_COMBINE, s, lv.n_elim);
        if (int rc = slv_launch_fold(ch, ch.d_fold + 3 * (size_t)sch.n_fold_pins, lv.n_elim, d_status, s)) return rc;
      }
      ch.AL0 = nullptr;
    }
    {
      const bool fused0 = level == 0 && ch.st != nullptr;
      const bool explicit_c = !fused0 && !(ch.implicit_couplings && lv.adjacent);
      // (measured and not kept in round 2: the first strip's operands requested BEFORE the factorisation and the other two
      //  together after it - the time moves into chol80 (12.8 -> 14-17 us under load), k_bcr_elim stays at 0.287 ms; a raised
      //  wave priority on the pivot chain and swapped-operand mirror tiles of G change nothing either.  Per-workgroup phase
      //  times under load, scripts/gpu_stamps.py: level 0 build 5.2 / chol80 14.0 / z + G 7.6 us; level 1 load 6 / chol80
      //  12.9 / strips 12-22 / store 1-3 us.  An EXTRA 51 KB store per node at levels 1-4 (+80 MB per step) costs 10 us:
      //  the wide eliminations are only mildly bandwidth-sensitive, so triangle-only transfers could save ~8 us at best.)
      // (64: measured again in round 2 - the strip form for levels of up to 128 / 256 / 512 nodes moves time from k_bcr_elim
      //  to k_bcr_elim_deep one for one: 0.889 / 0.902 / 0.968 ms per step against 0.893)
      // (128 since round 3: the separator chain of the chunked solver starts with ~119 eliminated nodes on an otherwise idle
      //  chip - two strip workgroups per node: 28.7 -> 25.4 us for that level)
      const bool deep = explicit_c && lv.n_elim <= 128;
      // (measured dead end, round 2: the wide levels >= 1 in the strip form of k_bcr_elim_deep with T = 1, compiled for
      //  THREE workgroups per CU - 168 VGPRs, 12 B/lane of scratch -: k_bcr_elim 0.288 -> 0.341 ms per step.  The
      //  level is not occupancy-bound; the third workgroup only adds contention for the one LDS pipe.)
      ProfSpan sp(prof, deep ? PC_ELIM_DEEP : PC_ELIM, s, lv.n_elim);
      if (deep) {   // narrow level: T workgroups per node
        // (isolated nodes of an incomplete reduction have no W strips: one workgroup per node factors and stores)
        const int T = lv.isolated ? 1 : std::min(10, 256 / lv.n_elim);   // strip workgroups per node (+ 1 that stores the factor)
        const int extra = (!lv.isolated && lv.n_elim * (T + 1) <= 256) ? 1 : 0;
        const int total = lv.n_elim * (T + extra), nx = std::min(8, (total + 31) / 32), per = (total + nx - 1) / nx;
        hipLaunchKernelGGL(k_bcr_elim_deep, dim3(8 * per), dim3(256), kElimDeepLds, s, ch,
                           ch.d_elim + 3 * lv.elim_off, d_numeric_err, d_status, T, extra, nx, per, total,
                           (level == 0 && add_al) ? ch.AL0 : (const double*)nullptr);
      } else
        hipLaunchKernelGGL(k_bcr_elim, dim3(lv.n_elim), dim3(256), kElimLds, s, ch, ch.d_elim + 3 * lv.elim_off,
                           d_c, d_numeric_err, d_status, level);
    }
    ACINO_LAUNCH_CHECK();
    if (dbg && !dbg_hit) dbg_hit = dbg_check("elim", level, ch, ch.d_elim + 3 * lv.elim_off, lv.n_elim, 3, d_numeric_err, s);
    if (lv.n_remain > 0) {
      {
        const bool up0 = level == 0 && ch.st != nullptr;
        ProfSpan sp(prof, up0 ? PC_UPDATE0 : (lv.n_remain <= 128 ? PC_UPDATE_DEEP : PC_UPDATE), s, lv.n_remain);
        if (up0)
          hipLaunchKernelGGL(k_bcr_update0, dim3(lv.n_remain), dim3(256), kUpdate0Lds, s, ch,
                             ch.d_remain + 4 * lv.remain_off, d_c, d_status);
        else if (lv.n_remain <= 128) {     // narrow level: <= 512 workgroups after the 2 S-way split
          const int S = lv.n_remain <= 32 ? 4 : (lv.n_remain <= 64 ? 2 : 1);
          const int total = (2 * S + 1) * lv.n_remain, nx = std::min(8, (total + 31) / 32), per = (total + nx - 1) / nx;
          hipLaunchKernelGGL(k_bcr_update_deep, dim3(8 * per), dim3(256), kUpdateDeepLds, s, ch,
                             ch.d_remain + 4 * lv.remain_off, d_status, S, nx, per, total,
                             (level == 0 && add_al) ? ch.AL0 : (const double*)nullptr);
        } else
          hipLaunchKernelGGL(k_bcr_update, dim3(3 * lv.n_remain), dim3(256), kUpdateLds, s, ch,
                             ch.d_remain + 4 * lv.remain_off, d_c, d_status, level);
      }
      ACINO_LAUNCH_CHECK();
      if (dbg && !dbg_hit) dbg_hit = dbg_check("update", level, ch, ch.d_remain + 4 * lv.remain_off, lv.n_remain, 4, d_numeric_err, s);
    }
    ++level;
  }
  if (fz && sch.n_fold_pins > 0) {       // the pins' Schur complements, for the export
    ProfSpan sp(prof, PC_SEP_COMBINE, s, sch.n_fold_pins);
    if (int rc = slv_launch_fold(ch_in, ch_in.d_fold, sch.n_fold_pins, d_status, s)) return rc;
  }
  if (!sch.pairs.empty() && ch.d_pairs && ch.trunc_eps2 && sch.refine == 0) {
    ProfSpan sp(prof, PC_TRUNC_CHECK, s, (long long)sch.pairs.size() / 2);
    hipLaunchKernelGGL(k_bcr_trunc_check, dim3((unsigned)(sch.pairs.size() / 2)), dim3(256), kTruncCheckLds, s, ch,
                       d_status);
    ACINO_LAUNCH_CHECK();
  }
  return ACINO_OK;
}

int bcr_backsub(const BcrChain& ch, const BcrSchedule& sch, const FteConst* d_c, const int* d_status, hipStream_t s,
                Profiler* prof, int* d_numeric_err) {
  int top = (int)sch.levels.size() - 1;
  if (sep_tail_applies(ch, sch)) {
    // one launch for the truncated solve, the sweeps and every level above them (k_sep_tail)
    const BcrLevel& iso = sch.levels[top];
    SepTailArgs a;
    a.iso = ch.d_elim + 3 * iso.elim_off;
    a.n_iso = iso.n_elim;
    a.refine = sch.refine;
    a.n_lv = top;
    int blocks = iso.n_elim;
    for (int k = top - 1, q = 0; k >= 0; --k, ++q) {
      a.lv_off[q] = sch.levels[k].elim_off;
      a.lv_cnt[q] = sch.levels[k].n_elim;
      blocks += sch.levels[k].n_elim;
    }
    a.ver = ch.st_flags;
    a.epoch = ch.st_flags + ch.n_st_flags;
    a.ll = ch.st_ll;
    a.norms = ch.trunc_eps2;
    a.fused = (sch.fused_levels && ch.SL != nullptr) ? 1 : 0;
    a.iso_loc = ch.d_iso_loc;
    {
      ProfSpan sp(prof, PC_REFINE, s, blocks);
      hipLaunchKernelGGL(k_sep_tail, dim3(blocks), dim3(ST_T), kBacksubTailLds, s, ch, a, d_status, d_numeric_err);
    }
    ACINO_LAUNCH_CHECK();
    return ACINO_OK;
  }
  if (sch.refine > 0 && top >= 0 && sch.levels[top].isolated && ch.refine_buf) {
    // truncated solve of the isolated nodes, then the block-Jacobi sweeps over their dropped couplings
    const BcrLevel& lv = sch.levels[top];
    const int* iso = ch.d_elim + 3 * lv.elim_off;
    {
      ProfSpan sp(prof, PC_BACKSUB, s, lv.n_elim);
      hipLaunchKernelGGL(k_bcr_backsub<512>, dim3(lv.n_elim), dim3(512), kBacksubLds, s, ch, iso, d_status);
    }
    ACINO_LAUNCH_CHECK();
    double* x0 = ch.refine_buf;
    double* it[2] = {x0 + (size_t)lv.n_elim * BS, x0 + 2 * (size_t)lv.n_elim * BS};
    for (int sw = 0; sw < sch.refine; ++sw) {
      const bool last = sw + 1 == sch.refine;
      ProfSpan sp(prof, PC_REFINE, s, lv.n_elim);
      // the last two and the first two sweeps record max |update| (and max |x|) per node: k_totals turns them into the bound
      // on the error.  trunc_eps2 layout with refinement, n = isolated nodes: [0, n) |update| and [n, 2n) |x| of the last
      // sweep, [2n, 3n) |update| and [3n, 4n) |x| of the sweep before it; with >= 3 sweeps [4n, 5n) |update| of the first,
      // with >= 4 [5n, 6n) |update| of the second (a sweep writes |x| behind its |update|: the first's is overwritten by the
      // second's |update|, the second's lands in [6n, 7n) and is not used)
      double* norms = last ? ch.trunc_eps2
                           : (sw + 2 == sch.refine ? ch.trunc_eps2 + 2 * (size_t)lv.n_elim
                                                   : (sw == 0 ? ch.trunc_eps2 + 4 * (size_t)lv.n_elim
                                                              : (sw == 1 ? ch.trunc_eps2 + 5 * (size_t)lv.n_elim : (double*)nullptr)));
      hipLaunchKernelGGL(k_bcr_refine, dim3(lv.n_elim), dim3(RF_T), kBacksubTailLds, s, ch, iso, lv.n_elim,
                         sw == 0 ? (const double*)nullptr : it[(sw - 1) & 1], it[sw & 1], x0, (last && sw > 0) ? 1 : 0, norms,
                         d_status);
      ACINO_LAUNCH_CHECK();
    }
    if (sch.refine == 1) {      // (a single sweep reads its neighbours from ch.b: the result goes there afterwards)
      hipLaunchKernelGGL(k_bcr_refine_copy, dim3(lv.n_elim), dim3(128), 0, s, ch, iso, it[0], d_status);
      ACINO_LAUNCH_CHECK();
    }
    --top;
  }
  if (ch.d_tail && sch.tail_levels > 0) {
    const int n_tail = (int)sch.tail.size() / 4;
    ACINO_HIP_CHECK(hipMemsetAsync(ch.d_done, 0, sizeof(int), s));
    {
      ProfSpan sp(prof, PC_BACKSUB_TAIL, s, n_tail);
      hipLaunchKernelGGL(k_bcr_backsub_tail, dim3(n_tail), dim3(256), kBacksubTailLds, s, ch, d_status, d_numeric_err);
    }
    ACINO_LAUNCH_CHECK();
    top -= sch.tail_levels;
  }
  for (int k = top; k >= 0; --k) {
    const BcrLevel& lv = sch.levels[k];
    {
      ProfSpan sp(prof, (k == 0 && ch.st != nullptr) ? PC_BACKSUB0 : PC_BACKSUB, s, lv.n_elim);
      if (k == 0 && ch.st != nullptr)
        hipLaunchKernelGGL(k_bcr_backsub0, dim3(lv.n_elim), dim3(256), 0, s, ch,
                           ch.d_elim + 3 * lv.elim_off, d_c, d_status);
      else
        if (lv.n_elim <= 256)
          hipLaunchKernelGGL(k_bcr_backsub<512>, dim3(lv.n_elim), dim3(512), kBacksubLds, s, ch,
                             ch.d_elim + 3 * lv.elim_off, d_status);
        else
          hipLaunchKernelGGL(k_bcr_backsub<256>, dim3(lv.n_elim), dim3(256), kBacksubLds, s, ch,
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

// Debug aid: fills the LDS of the CUs it lands on with NaN patterns (n_blocks workgroups x 64 KB).  Run on a second stream
// beside a solve it turns every read of LDS that the reading kernel did not write itself into a NaN.
namespace acino {
__global__ void __launch_bounds__(256) k_poison_lds(int spin) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* p = reinterpret_cast<double*>(smem_raw);
  for (int r = 0; r < spin; ++r)
    for (int e = threadIdx.x; e < 8192; e += 256) p[e] = __builtin_nan("");
  __syncthreads();
  if (p[threadIdx.x] == 1.0) p[0] = 2.0;     // keep the stores
}
}  // namespace acino
extern "C" int acino_debug_poison_lds(int n_blocks, int spin, void* stream) {
  using namespace acino;
  hipLaunchKernelGGL(k_poison_lds, dim3(n_blocks), dim3(256), 65536, (hipStream_t)stream, spin);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

extern "C" int acino_selftest_mfma(const double* d_a, const double* d_b, int k, double* d_c, void* stream) {
  using namespace acino;
  ACINO_REQUIRE(k > 0 && k % 4 == 0, "k must be a positive multiple of 4");
  ACINO_REQUIRE(d_a && d_b && d_c, "null buffer");
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, d_a, d_b, k, d_c);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}
