// Microbenchmark: the 16-pivot chain (chol16_inv_acc, dense80.hpp) on one wave - the latency every factorisation of this
// library is made of (5 per 80 x 80 node).  ns per chain, alone and with a quiet / busy SIMD mate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acinoset_amd/csrc -I include scripts/bench/chain16.hip -o /tmp/chain16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#include "dense80.hpp"
using namespace acino;
namespace acino {
template <int LDT = LD, bool KEEP_L = false>
__device__ __forceinline__ bool chol16_inv_acc_v2(double* T, d4 acc, int lane, int* err, double* Lout = nullptr) {
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
      const double pv = piv;
      if (k0 < 3) {
        // the NEXT pivot does not wait for 1/sqrt: a'(c+1,c+1) = a(c+1,c+1) - a(c+1,c)^2 / piv through the reciprocal
        // (estimate + one third-order step: rcp -> e -> e + e^2 -> r), two dependent operations fewer than via L[c+1][c]
        double r = __builtin_amdgcn_rcp(pv);
        const double er = fma(-pv, r, 1.0);
        r = fma(r, fma(er, er, er), r);
        piv = fma(-(mraw[k0 + 1] * mraw[k0 + 1]), r, dnext);
      }
      y = __builtin_amdgcn_rsq(pv);
      const double e = fma(-(pv * y), y, 1.0);
      y = fma(y * e, fma(e, 0.375, 0.5), y);
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
}
template <int V>
__global__ void __launch_bounds__(512) k_chain(long long* out, double* sink, int reps, int mode, double* tile_out) {
  __shared__ double T[16 * LD], T0[16 * LD];
  __shared__ int err;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  // an SPD tile: diagonally dominant
  for (int e = tid; e < 16 * 16; e += blockDim.x) {
    const int r = e / 16, c = e % 16;
    T0[r * LD + c] = (r == c ? 20.0 : 0.0) + 0.3 * ((r * 7 + c * 3) % 5) + 0.3 * ((c * 7 + r * 3) % 5);
  }
  if (tid == 0) err = 0;
  __syncthreads();
  long long t = 0;
  double acc_sink = 0.0;
  if (wave == 0) {
    const long long w0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
      d4 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = T0[(lk + 4 * q) * LD + li];
      if (V == 0) chol16_inv_acc(T, acc, lane, &err); else chol16_inv_acc_v2(T, acc, lane, &err);
      acc_sink += T[li * LD + lk];
    }
    t = wall_clock64() - w0;
  } else if (mode == 1 && wave == 4) {        // the SIMD mate streams matrix instructions
    d4 a = {0, 0, 0, 0};
    for (int r = 0; r < reps * 20; ++r) a = mfma(1e-3 * lane, 1.0, a);
    acc_sink = a[0];
  } else if (mode == 2 && wave == 4) {        // the SIMD mate streams vector instructions
    double a = lane;
    for (int r = 0; r < reps * 200; ++r) a = fma(a, 1.0000001, 1e-9);
    acc_sink = a;
  }
  if (tid == 0) out[blockIdx.x] = t;
  __syncthreads();
  if (blockIdx.x == 0 && tile_out) for (int e = tid; e < 256; e += blockDim.x) tile_out[e] = T[(e / 16) * LD + e % 16];
  if (acc_sink == 12345.678) sink[tid] = acc_sink;
}
int main() {
  long long* d; double* sink;
  hipMalloc(&d, 8 * 256); hipMalloc(&sink, 8 * 1024);
  const int reps = 400;
  const char* names[] = {"quiet SIMD mate", "SIMD mate streams fp64 matrix instructions", "SIMD mate streams fp64 FMAs"};
  double* tiles; hipMalloc(&tiles, 8 * 512);
  for (int v = 0; v < 2; ++v)
  for (int blocks : {1, 256})
    for (int mode = 0; mode < 3; ++mode) {
      for (int it = 0; it < 2; ++it) {
        if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(512), 0, 0, d, sink, reps, mode, tiles);
        else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(512), 0, 0, d, sink, reps, mode, tiles + 256);
      }
      hipDeviceSynchronize();
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
      long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      printf("16-pivot chain v%d, %3d workgroups, %-44s: %7.1f ns per chain\n", v, blocks, names[mode], 10.0 * mx / reps);
    }
  std::vector<double> ht(512);
  hipMemcpy(ht.data(), tiles, 8 * 512, hipMemcpyDeviceToHost);
  double md = 0, mv = 0;
  for (int e = 0; e < 256; ++e) { md = std::max(md, std::abs(ht[e] - ht[256 + e])); mv = std::max(mv, std::abs(ht[e])); }
  printf("max |U_v0 - U_v1| = %.3e (max |U| %.3e)\n", md, mv);
  return 0;
}
