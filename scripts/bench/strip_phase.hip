// Microbenchmark of k_sep_level's strip phase (W = U^T [C_l | C_r] in place, 512 threads, two strips per round) on one CU or on
// all of them: where do ~2 us per round go when the matrix instructions of a round are ~1 us per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acinoset_amd/csrc -I include scripts/bench/strip_phase.hip -o /tmp/strip_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dense80.hpp"
using namespace acino;

template <int IB>
__device__ __forceinline__ void row_tile(const double* Lm, const double (&bv)[20], double* Wb, int cc, int li, int lk, bool do_mfma) {
  constexpr int NS = 4 * (IB + 1);
  double av[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) av[t] = Lm[(4 * t + lk) * LD + IB * 16 + li];
  d4 acc = {0, 0, 0, 0};
  if (do_mfma) {
#pragma unroll
    for (int t = 0; t < NS; ++t) acc = mfma(av[t], bv[t], acc);
  } else {
#pragma unroll
    for (int t = 0; t < NS; ++t) acc[t & 3] += av[t] * bv[t];
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) Wb[(IB * 16 + lk + 4 * rr) * LD + cc + li] = acc[rr];
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// mode 0: as in the kernel; 1: no matrix instructions (FMA stand-ins); 2: no barriers (results wrong, timing only);
// 3: roles not mirrored; 4: one strip per round on waves 0..3 only
__global__ void __launch_bounds__(512) k_strips(long long* out, double* sink, int ns, int mode, int reps) {
  extern __shared__ double lds[];
  double* Lm = lds;
  double* WL = Lm + MAT;
  double* WR = WL + MAT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 3 * MAT; e += 512) lds[e] = 1e-3 * ((e * 7) % 97) - 0.04;
  __syncthreads();
  const long long w0 = wall_clock64();
  for (int rep = 0; rep < reps; ++rep) {
    const int per = mode == 4 ? 1 : 2;
    for (int q0 = 0; q0 < ns; q0 += per) {
      const int q = q0 + (mode == 4 ? 0 : (wave >> 2));
      const bool on = q < ns && (mode != 4 || wave < 4);
      const int s = q < ns ? q : 0, side = s >= 5, cc = 16 * (side ? s - 5 : s);
      double* Wb = side ? WR : WL;
      double bv[20];
#pragma unroll
      for (int t = 0; t < 20; ++t) bv[t] = Wb[(4 * t + lk) * LD + cc + li];
      if (mode != 2) lds_barrier();
      if (on) {
        const int role = (wave < 4 || mode == 3) ? (wave & 3) : 7 - wave;
        if (role == 0) row_tile<4>(Lm, bv, Wb, cc, li, lk, mode != 1);
        else if (role == 1) row_tile<3>(Lm, bv, Wb, cc, li, lk, mode != 1);
        else if (role == 2) {
          row_tile<2>(Lm, bv, Wb, cc, li, lk, mode != 1);
          row_tile<0>(Lm, bv, Wb, cc, li, lk, mode != 1);
        } else row_tile<1>(Lm, bv, Wb, cc, li, lk, mode != 1);
      }
    }
    if (mode != 2) lds_barrier();
  }
  const long long w1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = w1 - w0;
  if (WL[tid] == 12345.678) sink[tid] = WL[tid];
}
// the products phase: nt tiles over 8 waves, 20 matrix instructions each, both operands from LDS (column pattern)
__global__ void __launch_bounds__(512) k_products(long long* out, double* sink, int nt, int mode, int reps) {
  extern __shared__ double lds[];
  double* WL = lds + MAT;
  double* WR = WL + MAT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 3 * MAT; e += 512) lds[e] = 1e-3 * ((e * 7) % 97) - 0.04;
  __syncthreads();
  const long long w0 = wall_clock64();
  d4 tot = {0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    for (int q = wave; q < nt; q += 8) {
      const int ta = (q / 5) % 5, tb = q % 5;
      const double* A = (q & 1) ? WR : WL;
      d4 acc = mma_seq<BS / 4, true>(d4{0, 0, 0, 0}, A + lk * LD + ta * 16 + li, 4 * LD, WL + lk * LD + tb * 16 + li, 4 * LD);
      tot += acc;
    }
  }
  const long long w1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = w1 - w0;
  if (tot[0] + tot[1] + tot[2] + tot[3] == 12345.678) sink[tid] = tot[0];
}
int main() {
  long long* d; double* sink;
  hipMalloc(&d, 8 * 512); hipMalloc(&sink, 8 * 1024);
  const size_t lds = 3 * MAT * sizeof(double);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_strips), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_products), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int reps = 50;
  const char* names[] = {"as in the kernel", "no matrix instructions (FMA stand-ins)", "no barriers", "roles not mirrored", "one strip per round, 4 waves"};
  for (int blocks : {1, 238}) {
    for (int mode = 0; mode < 5; ++mode) {
      for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_strips, dim3(blocks), dim3(512), lds, 0, d, sink, 10, mode, reps);
      hipDeviceSynchronize();
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
      long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      printf("strips, 10 strips, %3d workgroups, %-42s: %6.2f us per phase (slowest workgroup)\n", blocks, names[mode], mx / 100.0 / reps);
    }
    for (int nt : {25, 30}) {
      for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_products, dim3(blocks), dim3(512), lds, 0, d, sink, nt, 0, reps);
      hipDeviceSynchronize();
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
      long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      printf("products, %d tiles, %3d workgroups: %6.2f us per phase\n", nt, blocks, mx / 100.0 / reps);
    }
  }
  return 0;
}
