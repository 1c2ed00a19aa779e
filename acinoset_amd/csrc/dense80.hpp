// Dense 80 x 80 fp64 block machinery shared by the block cyclic reduction (bcr.hip) and the Kalman smoother
// (ekf.hip): 16 x 16 tiles on the matrix cores (v_mfma_f64_16x16x4_f64), the register-resident 16 x 16 Cholesky with
// its inverse factor, the blocked 80 x 80 Cholesky built on it, and the HBM <-> LDS block moves.  LDS leading dimension
// 81 makes both the row-pattern and the column-pattern MFMA operand reads bank-conflict free.  256 threads (four waves).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fte_kernels.hpp"

namespace acino {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int LD = 81;
constexpr int MAT = BS * LD;
constexpr int NT = 5;

// Workgroups are dealt to the 8 XCDs round-robin (workgroup k runs on XCD k % 8) and every XCD has its own L2.  Kernels
// whose NEIGHBOURING work items read the same 51 KB operand (the two remaining neighbours of an eliminated node, the
// two roles of an update) map the hardware index to a logical index such that each XCD works on one contiguous range:
// the second reader then finds the operand in its L2 instead of fetching it again through the fabric.
__device__ __forceinline__ int xcd_contiguous(int k, int n) {
  const int x = k & 7, q = k >> 3, base = n >> 3, rem = n & 7;
  return x * base + (x < rem ? x : rem) + q;
}

// The same contiguous ranges walked BACKWARDS: a consumer kernel that starts with the work items whose operands the
// producer kernel (same mapping, forwards) wrote last finds them still in the XCD's L2.
__device__ __forceinline__ int xcd_contiguous_rev(int k, int n) {
  const int x = k & 7, q = k >> 3, base = n >> 3, rem = n & 7;
  const int cnt = base + (x < rem ? 1 : 0);
  return x * base + (x < rem ? x : rem) + (cnt - 1 - q);
}

__device__ __forceinline__ double readlane_d(double x, int lane) {
  long long b = __builtin_bit_cast(long long, x);
  int lo = __builtin_amdgcn_readlane((int)b, lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// gfx950 row swaps used below: v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows
// of the second, v_permlane32_swap the upper half of the first with the lower half of the second - plain VALU latency
// instead of trips through the LDS crossbar (ds_bpermute).  The builtins return {new first, new second}.
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// NS back-to-back MFMAs on one accumulator with ALL operand reads issued first (software pipelining:
// an LDS read costs ~100+ cycles of latency, an fp64 MFMA 64 cycles of issue).  pa/pb are this lane's
// operand pointers for k-step 0; k-step s reads pa[s*sa], pb[s*sb].
template <int NS, bool NEG>
__device__ __forceinline__ d4 mma_seq(d4 acc, const double* pa, int sa, const double* pb, int sb) {
  double av[NS], bv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    av[s] = pa[s * sa];
    bv[s] = pb[s * sb];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) acc = mfma(NEG ? -av[s] : av[s], bv[s], acc);
  return acc;
}

// Panel gather for the 16-pivot chain.  Lane (i, k) holds a = C[i][4s+k] and u = Uwork[i][4s+k]; afterwards the lanes of
// rows k = 0, 1 hold x[j] = C[i][4s+j] and the lanes of rows k = 2, 3 hold x[j] = Uwork[i][4s+j], j = 0..3 - both work
// matrices of the panel side by side in ONE register set, so every column operation of the chain is one instruction
// for both.  Three gfx950 row swaps per 32-bit half (semantics above).
__device__ __forceinline__ void panel_gather(double a, double u, double (&x)[4]) {
  const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, u);
  const u2v sl = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)ub, false, false);                 // [0]: a0 a1 u0 u1   [1]: a2 a3 u2 u3
  const u2v sh = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(ub >> 32), false, false);
  const u2v l01 = __builtin_amdgcn_permlane16_swap(sl[0], sl[0], false, false);   // [0]: a0 a0 u0 u0   [1]: a1 a1 u1 u1
  const u2v l23 = __builtin_amdgcn_permlane16_swap(sl[1], sl[1], false, false);   // [0]: a2 a2 u2 u2   [1]: a3 a3 u3 u3
  const u2v h01 = __builtin_amdgcn_permlane16_swap(sh[0], sh[0], false, false);
  const u2v h23 = __builtin_amdgcn_permlane16_swap(sh[1], sh[1], false, false);
  x[0] = __builtin_bit_cast(double, ((unsigned long long)h01[0] << 32) | l01[0]);
  x[1] = __builtin_bit_cast(double, ((unsigned long long)h01[1] << 32) | l01[1]);
  x[2] = __builtin_bit_cast(double, ((unsigned long long)h23[0] << 32) | l23[0]);
  x[3] = __builtin_bit_cast(double, ((unsigned long long)h23[1] << 32) | l23[1]);
}
// The value of v held by lane (i, k ^ 2): the two halves of the wave trade places.
__device__ __forceinline__ double half_swap(double v, bool upper) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const u2v l = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);          // [0]: lo lo   [1]: hi hi
  const u2v h = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  const unsigned lo = upper ? l[0] : l[1], hi = upper ? h[0] : h[1];
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// One wave (all 64 lanes): Cholesky of the symmetric 16x16 tile T (LDS, leading dim LD) and the inverse of
// its factor, entirely in registers.  Lane (i = lane & 15, k = lane >> 4) holds row i, columns {k, k+4, k+8,
// k+12}: acc[r] = C[i][4r+k] - which IS the MFMA C layout of the symmetric tile.  The right-looking column
// operations that turn C into L are applied at the same time to an identity tile (uacc), which they turn into
// U = L^-T: no separate triangular inversion.  Columns go in four panels of four.  The chain wave is bound by VALU
// ISSUE (it keeps its SIMD's port ~65 % busy), so the panel is laid out to need the fewest
// instructions: panel_gather puts the C columns into the lower half of the wave and the identity columns into the upper
// half, one register set x[0..3] for both, and every scale / eliminate step is ONE instruction for the two matrices.
// The pivots and multipliers L[4s+k'][c] are wave-uniform and travel through SGPRs (v_readlane from the lower half), so
// the 16-pivot dependency chain is readlane -> rsq -> Newton -> mul -> readlane -> fma; for the rank-4 update of the
// remaining columns the halves trade the two columns the other one lacks (half_swap) and ONE v_mfma_f64_16x16x4 per
// tile applies it (the finished panel register is already both the A and the B operand).  The tile is OVERWRITTEN by
// U_kk (upper triangular); L_kk is not kept.  No per-pivot checks: a non-positive (or NaN) pivot turns 1/sqrt into
// NaN / inf, which the column operations and the rank-4 updates carry into every later pivot - ONE test of the last
// 1/sqrt flags the tile.
// LDT: leading dimension of T (and of Lout).  With KEEP_L the factor itself goes to Lout (lower triangle, zeros above).
// Returns true when every pivot was positive (wave-uniform).
template <int LDT = LD, bool KEEP_L = false>
__device__ __forceinline__ bool chol16_inv_acc(double* T, d4 acc, int lane, int* err, double* Lout = nullptr) {
  const int i = lane & 15, k = lane >> 4;
  const bool upper = k >= 2, odd = k & 1;
  d4 uacc;
#pragma unroll
  for (int r = 0; r < 4; ++r) uacc[r] = (i == 4 * r + k) ? 1.0 : 0.0;
  double y = 1.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double x[4];
    panel_gather(acc[s], uacc[s], x);
    // The pivot chain inside the panel is kept as short as the arithmetic allows: the multipliers and the next
    // diagonal entry are broadcast RAW (before this pivot's 1/sqrt is known, i.e. beside its rsq chain), and the next
    // pivot a'(c+1,c+1) - (a(c+1,c) y)^2 is formed directly from them: rsq -> Newton -> mul -> fma -> next rsq.
    double piv = readlane_d(x[0], 4 * s);
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
      double mraw[4] = {0, 0, 0, 0};
#pragma unroll
      for (int kk = k0 + 1; kk < 4; ++kk) mraw[kk] = readlane_d(x[k0], 4 * s + kk);      // a[4s+kk][c], unscaled
      const double dnext = k0 < 3 ? readlane_d(x[k0 + 1], 4 * s + k0 + 1) : 0.0;         // a[c+1][c+1] so far
      // 1/sqrt(piv): hardware estimate + one coupled Newton step (no range fix-ups: piv is a positive normal number)
      y = __builtin_amdgcn_rsq(piv);
      const double e = fma(-(piv * y), y, 1.0);
      y = fma(y * e, fma(e, 0.375, 0.5), y);
      if (k0 < 3) {
        const double t = mraw[k0 + 1] * y;                    // L[c+1][c]
        piv = fma(-t, t, dnext);
      }
      x[k0] *= y;                                             // row c becomes sqrt(piv); rows < c hold don't-cares
#pragma unroll
      for (int kk = k0 + 1; kk < 4; ++kk) x[kk] -= x[k0] * (mraw[kk] * y);   // (mraw * y) = L[4s+kk][c]
    }
    if (k == 2) {                                             // U[i][4s .. 4s+3] is final
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) T[i * LDT + 4 * s + kk] = x[kk];
    }
    if (KEEP_L && k == 1) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) Lout[i * LDT + 4 * s + kk] = (i >= 4 * s + kk) ? x[kk] : 0.0;
    }
    if (s < 3) {
      const double xa = odd ? x[1] : x[0], xb = odd ? x[3] : x[2];
      const double mine = upper ? xb : xa;                    // x[k]:      L column k (lower half) / U column k (upper half)
      const double other = half_swap(upper ? xa : xb, upper); // x[k ^ 2] of lane (i, k ^ 2): the column this half lacks
      const double pk = upper ? other : mine, pu = upper ? mine : other;
      const double p = (i >= 4 * s + k) ? pk : 0.0;           // strictly-upper entries are discarded here, once
      acc = mfma(-p, p, acc);
      uacc = mfma(-p, pu, uacc);     // register r of lane (i,k) is result[4r+k][i] = -(P PU^T)[4r+k][i] = dU[i][4r+k]
    }
  }
  const bool bad = !(fabs(y) < 1e300);                        // NaN or inf (wave-uniform)
  if (bad && err && lane == 0) atomicOr(err, 1);
  return !bad;
}
__device__ __forceinline__ void chol16_inv(double* T, int lane, int* err) {
  const int i = lane & 15, k = lane >> 4;
  d4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = T[(k + 4 * r) * LD + i];
  chol16_inv_acc(T, acc, lane, err);
}

// Blocked Cholesky of the 80x80 matrix in LDS, with the inverse factor built alongside.  On exit: strictly-lower
// tiles hold L(ib,jb), diagonal tiles U_kk = L_kk^-T and strictly-upper tiles (j,c) hold U(j,c), U = L^-T
// (80x80 upper triangular).  The column operations that reduce A to L are applied at block level to the identity as
// well ([A; I] L^-T = [L; L^-T]); the work tiles of that second matrix live in the (otherwise unused) strictly-upper
// tiles, so nothing is zero-filled: tile (j,c) is first WRITTEN at step kb = j and accumulated afterwards.
// Per block column kb:  panel  tile(t,kb) <- tile(t,kb) U_kk for every t != kb (4 tiles, one per wave);
//                       trailing tile(t,c) -= tile(t,kb) L(c,kb)^T for c > kb, t in {0..kb} u {c..4}.
// LOOK-AHEAD: in the trailing phase wave 0 takes only the next diagonal tile, keeps the result in registers (the
// MFMA C layout is the factorisation's layout) and goes straight into its 16-pivot chain, while waves 1..3 do all
// other trailing tiles (<= 5 each) - the inverse costs no time on the critical path.
// Task tables as ARITHMETIC on packed literals: a __constant__ byte table costs a global_load + s_waitcnt vmcnt(0) at every
// lookup (gfx950 has no sub-dword scalar loads) - on the pivot chain's critical path that was ~0.5 us per block column.
// trailing tasks of waves 1..3 at block column kb: code = 16*ti + tj ; ti <= kb marks a tile of U (overwrite when ti == kb)
//   kb 0: 21 22 31 32 33 41 42 43 44 01 02 03 04 | kb 1: 32 33 42 43 44 02 03 04 12 13 14 | kb 2: 43 44 03 04 13 14 23 24
//   kb 3: 04 14 24 34
__device__ __forceinline__ int trail_n(int kb) { return (0x04080B0Du >> (8 * kb)) & 0xFF; }
__device__ __forceinline__ int trail_code(int kb, int t) {
  const unsigned long long lo = kb == 0 ? 0x4342413332312221ull
                                : (kb == 1 ? 0x0403024443423332ull : (kb == 2 ? 0x2423141304034443ull : 0x34241404ull));
  const unsigned long long hi = kb == 0 ? 0x0403020144ull : (kb == 1 ? 0x141312ull : 0ull);
  return (int)(((t < 8 ? lo : hi) >> (8 * (t & 7))) & 0xFF);
}
// panel: row tile q of block column kb, {1,2,3,4}, {2,0,3,4}, {3,0,1,4}, {4,0,1,2}, {0,1,2,3}: entry 0 is always the tile
// the look-ahead needs next
__device__ __forceinline__ int panel_tile(int kb, int q) {
  const unsigned v = kb < 4 ? (unsigned)(0x2104410343024321ull >> (16 * kb)) & 0xFFFFu : 0x3210u;
  return (v >> (4 * q)) & 15;
}

// NW: waves of the workgroup (4, or 8 in the 512-thread kernels: the extra waves share the trailing tiles)
struct Chol80NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
// hook(kb): called by every wave but the pivot chain's once column block kb of U is final (after its trailing tiles of step kb;
// for the last block after the last panel) - work that only needs the finished columns rides under the chain of block kb + 1
template <int NW = 4, class Hook = Chol80NoHook>
__device__ __forceinline__ void chol80(double* Lm, int tid, int* err, long long* dbg = nullptr, Hook&& hook = Hook()) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  if (wave == 0) chol16_inv(Lm, lane, err);
  __syncthreads();
  for (int kb = 0; kb < NT; ++kb) {
    const double* Ukk = Lm + (kb * 16) * LD + kb * 16;
    if (NW == 4 || wave < 4) {  // panel: tile(t,kb) <- tile(t,kb) * U_kk   (L(ib,kb) = A(ib,kb) L_kk^-T below, U(j,kb) above the diagonal)
      double* A = Lm + (panel_tile(kb, wave) * 16) * LD + kb * 16;
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
    if (kb == NT - 1) {
      if (wave != 0) hook(kb);
      break;
    }
    if (wave == 0) {
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
      if (dbg && kb == 0 && tid == 0) dbg[16] = (long long)wall_clock64();
      chol16_inv_acc(Cc, a, lane, err);
      if (dbg && kb == 0 && tid == 0) dbg[17] = (long long)wall_clock64();
    } else {
      const int ntask = trail_n(kb);
      // (NW = 8: wave 4 shares the pivot chain's SIMD - an fp64 matrix instruction there holds the chain's issue port for 64
      //  cycles - and stays out; the six waves of the other three SIMDs take the tiles)
      const int helper = NW == 8 ? (wave < 4 ? wave - 1 : wave - 2) : wave - 1, n_help = NW == 8 ? 6 : NW - 1;
      for (int t = (NW == 8 && wave == 4) ? ntask : helper; t < ntask; t += n_help) {
        const int code = trail_code(kb, t), ti = code >> 4, tj = code & 15;
        double* Cc = Lm + (ti * 16) * LD + tj * 16;
        const double* A = Lm + (ti * 16) * LD + kb * 16;
        const double* B = Lm + (tj * 16) * LD + kb * 16;
        d4 a = {0, 0, 0, 0};
        double av[4], bv[4];
        if (ti != kb) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(lk + 4 * rr) * LD + li];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          av[s] = A[li * LD + 4 * s + lk];
          bv[s] = B[li * LD + 4 * s + lk];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma(-av[s], bv[s], a);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cc[(lk + 4 * rr) * LD + li] = a[rr];
      }
      hook(kb);
    }
    __syncthreads();
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
// The two halves of load_mat, for kernels that want several matrices in flight but only ONE LDS buffer: request a
// matrix into registers now (13 x 16 B per thread), stage it into LDS when its turn comes.
template <int NTH = 256>
__device__ __forceinline__ void fetch_mat(double2 (&v)[(BS * BS / 2 + NTH - 1) / NTH], const double* __restrict__ src, int tid) {
  const double2* s2 = reinterpret_cast<const double2*>(src);
#pragma unroll
  for (int k = 0; k < (BS * BS / 2 + NTH - 1) / NTH; ++k) {
    const int idx = tid + NTH * k;
    if (idx < BS * BS / 2) v[k] = s2[idx];
  }
}
template <int NTH = 256>
__device__ __forceinline__ void stage_mat(double* dst, const double2 (&v)[(BS * BS / 2 + NTH - 1) / NTH], int tid) {
#pragma unroll
  for (int k = 0; k < (BS * BS / 2 + NTH - 1) / NTH; ++k) {
    const int idx = tid + NTH * k;
    if (idx < BS * BS / 2) {
      const int e = 2 * idx, r = e / BS, c = e % BS;
      dst[r * LD + c] = v[k].x;
      dst[r * LD + c + 1] = v[k].y;
    }
  }
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

// Two 80x80 matrices HBM -> LDS with ALL loads of both in flight before the first LDS write (the narrow levels
// of the reduction are latency-bound: one HBM round trip instead of two).
__device__ __forceinline__ void load_mat2(double* dst0, const double* __restrict__ src0, double* dst1,
                                          const double* __restrict__ src1, int tid) {
  const double2* s0 = reinterpret_cast<const double2*>(src0);
  const double2* s1 = reinterpret_cast<const double2*>(src1);
  double2 v0[13], v1[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int idx = tid + 256 * k;
    if (idx < BS * BS / 2) {
      v0[k] = s0[idx];
      v1[k] = s1[idx];
    }
  }
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int idx = tid + 256 * k;
    if (idx < BS * BS / 2) {
      const int e = 2 * idx, r = e / BS, c = e % BS;
      dst0[r * LD + c] = v0[k].x;
      dst0[r * LD + c + 1] = v0[k].y;
      dst1[r * LD + c] = v1[k].x;
      dst1[r * LD + c + 1] = v1[k].y;
    }
  }
}

}  // namespace acino
