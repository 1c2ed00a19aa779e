// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on one SIMD (what a "matrix instruction" costs in the sweep's budgets).
//   modes: NACC independent accumulators, operands from registers | from LDS (one ds_read_b64 per instruction, pipelined 2 ahead)
//   waves per SIMD 1 or 2 (256 / 512 threads), one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma64_rate.hip -o /tmp/mfma64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int NACC, bool LDSOP>
__global__ void k_rate(long long* out, double* sink, int reps) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 80 * 81; e += blockDim.x) lds[e] = 1e-3 * (e % 97);
  __syncthreads();
  d4 acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = d4{0, 0, 0, 0};
  double x = 1e-3 * lane, y = 1.0 + 1e-9 * lane;
  const double* p = lds + (lane >> 4) * 81 + (lane & 15);
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    if (LDSOP) {
      double a[3][NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q) { a[0][q] = p[q * 16]; a[1][q] = p[4 * 81 + q * 16]; }
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        if (st + 2 < 16) {
#pragma unroll
          for (int q = 0; q < NACC; ++q) a[(st + 2) % 3][q] = p[(4 * (st + 2)) * 81 + (q % 5) * 16];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = mfma(a[st % 3][q], y, acc[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = mfma(x, y, acc[q]);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  if (s == 12345.678) sink[tid] = s;
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = c1 - c0; out[(blockIdx.x * 8 + wave) * 2 + 1] = w1 - w0; }
}

template <int NACC, bool LDSOP>
void run(const char* name, int threads, int blocks) {
  long long* d; double* sink;
  hipMalloc(&d, sizeof(long long) * blocks * 16); hipMalloc(&sink, 8 * 1024);
  const int reps = 200;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<NACC, LDSOP>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_rate<NACC, LDSOP>), dim3(blocks), dim3(threads), 150000, 0, d, sink, reps);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * 16);
  hipMemcpy(h.data(), d, sizeof(long long) * blocks * 16, hipMemcpyDeviceToHost);
  const double n = 16.0 * NACC * reps;
  printf("%-46s %3d thr x %3d wg: shader cycles per instruction, waves of wg 0:", name, threads, blocks);
  for (int w = 0; w < threads / 64; ++w) printf(" %6.1f", h[2 * w] / n);
  long long wmax = 0;
  for (int w = 0; w < threads / 64; ++w) wmax = std::max(wmax, h[2 * w + 1]);
  printf("  | slowest wave: %6.2f ns per instruction; per SIMD %6.2f ns\n", 10.0 * wmax / n, 10.0 * wmax / n / (threads / 256));
  hipFree(d); hipFree(sink);
}
int main() {
  for (int blocks : {1, 256}) {
    run<1, false>("1 accumulator (dependent), registers", 256, blocks);
    run<4, false>("4 accumulators, registers", 256, blocks);
    run<5, false>("5 accumulators, registers", 256, blocks);
    run<5, false>("5 accumulators, registers", 512, blocks);
    run<5, true>("5 accumulators, A operand from LDS", 256, blocks);
    run<5, true>("5 accumulators, A operand from LDS", 512, blocks);
    run<3, true>("3 accumulators, A operand from LDS", 512, blocks);
  }
  return 0;
}
