// Microbenchmark: what slows the register-resident 16x16 pivot chain (dense80.hpp: chol16_inv_acc) when other waves
// share its CU?  Wave 0 of every workgroup runs the chain REPS times; waves 1..3 run one kind of filler until wave 0
// is done.  Workgroups per CU are set through the dynamic LDS size.  Prints ns and shader cycles per chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I acinoset_amd/csrc scripts/contention_probe.hip -o /tmp/contention_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "dense80.hpp"

using namespace acino;

enum { IDLE = 0, MFMA64, VALU64, LDSRD, CHAIN, VALU32, SCALAR, BARRIER, VMEM, MFMADEP, TRAIL, NOP2, NOP3, NOP4, NOP5, NMODES };
static const char* kNames[NMODES] = {"idle (waves 1-3 wait at the barrier)", "fp64 MFMA 16x16x4, back to back", "fp64 VALU fma, 4 chains",
                                     "LDS reads (ds_read_b64)", "every wave runs its own pivot chain", "fp32 VALU fma, 4 chains",
                                     "v_readlane + SALU", "s_barrier-free spin on LDS flag only", "global loads (L2 hits)",
                                     "fp64 MFMA, ONE dependent accumulator chain", "chol80 trailing-tile loop (LDS -> 4 MFMA -> LDS)",
                                     "dependent MFMA + 2 x s_nop 15", "dependent MFMA + 3 x s_nop 15", "dependent MFMA + 4 x s_nop 15", "dependent MFMA + 5 x s_nop 15"};

template <int MODE, bool PRIO = false>
__global__ void __launch_bounds__(256) k_probe(const double* __restrict__ A, long long* out, int reps, double* sink) {
  extern __shared__ double lds[];
  __shared__ int stop;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, k = lane >> 4;
  double* T = lds + wave * 16 * LD;
  if (tid == 0) stop = 0;
  __syncthreads();
  double keep = 0.0;
  if (wave == 0 || MODE == CHAIN) {
    d4 a0;
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = A[(k + 4 * r) * 16 + i];
    T[i * LD + k] = 0.0;
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int r = 0; r < reps; ++r) {
      d4 a = a0;
      a[0] += 1e-280 * T[i * LD + k];       // depends on the previous factor: nothing can be hoisted
      chol16_inv_acc(T, a, lane, nullptr);
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (lane == 0) {
      out[(blockIdx.x * 4 + wave) * 2 + 0] = w1 - w0;
      out[(blockIdx.x * 4 + wave) * 2 + 1] = c1 - c0;
    }
    keep = T[i * LD + k];
    if (wave == 0 && lane == 0) __atomic_store_n(&stop, 1, __ATOMIC_RELAXED);
  } else if (MODE == MFMA64) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double x = 1e-3 * lane, y = 1.0 + 1e-9 * lane;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c0 = mfma(x, y, c0); c1 = mfma(y, x, c1); c2 = mfma(x, x, c2); c3 = mfma(y, y, c3);
      }
    }
    keep = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (MODE == VALU64) {
    double p = 1e-3 * lane, q = 2e-3 * lane, r = 3e-3 * lane, s = 4e-3 * lane;
    const double m = 1.0 - 1e-12, b = 1e-13;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { p = fma(p, m, b); q = fma(q, m, b); r = fma(r, m, b); s = fma(s, m, b); }
    }
    keep = p + q + r + s;
  } else if (MODE == VALU32) {
    float p = 1e-3f * lane, q = 2e-3f * lane, r = 3e-3f * lane, s = 4e-3f * lane;
    const float m = 0.999999f, b = 1e-7f;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { p = fmaf(p, m, b); q = fmaf(q, m, b); r = fmaf(r, m, b); s = fmaf(s, m, b); }
    }
    keep = (double)(p + q + r + s);
  } else if (MODE == LDSRD) {
    double acc = 0.0;
    T[lane] = 1.0 * lane;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += ((volatile double*)T)[(lane + u) & 63];
    }
    keep = acc;
  } else if (MODE == SCALAR) {
    int v = lane, acc = 0;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { acc += __builtin_amdgcn_readlane(v, u) * 3 + 1; v ^= acc; }
    }
    keep = (double)(acc + v);
  } else if (MODE == VMEM) {
    double acc = 0.0;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += ((const volatile double*)A)[(lane + 64 * u) & 255];
    }
    keep = acc;
  } else if (MODE == MFMADEP) {
    long long nd = 0;
    d4 c0 = {0, 0, 0, 0};
    const double x = 1e-3 * lane, y = 1.0 + 1e-9 * lane;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int q = 0; q < 16; ++q) c0 = mfma(x, y, c0);
      nd += 16;
    }
    keep = c0[0];
    if (lane == 0) out[(blockIdx.x * 4 + wave) * 2] = nd;
  } else if (MODE >= NOP2 && MODE <= NOP5) {
    d4 c0 = {0, 0, 0, 0};
    const double x = 1e-3 * lane, y = 1.0 + 1e-9 * lane;
    long long n = 0;
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        c0 = mfma(x, y, c0);
#pragma unroll
        for (int z = 0; z < MODE - NOP2 + 2; ++z) asm volatile("s_nop 15");
      }
      n += 16;
    }
    keep = c0[0];
    if (lane == 0) out[(blockIdx.x * 4 + wave) * 2] = n;
  } else if (MODE == TRAIL) {
    double* Lm = lds + 16 * LD;             // rows 16.. of an 80 x 81 matrix whose first tile row belongs to the chain
    Lm -= 16 * LD;
    for (int e = tid - 64; e < 64 * LD; e += 192) lds[16 * LD + e] = 1e-3 * (e % 97);
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) {
      for (int t = wave - 1; t < 9; t += 3) {               // the nine lower trailing tiles of block column 0
        const int code = c_trail[0][t], ti = code >> 4, tj = code & 15;
        double* Cc = Lm + (ti * 16) * LD + tj * 16;
        const double* Am = Lm + (ti * 16) * LD;
        const double* Bm = Lm + (tj * 16) * LD;
        d4 a; double av[4], bv[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a[rr] = Cc[(k + 4 * rr) * LD + i];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = Am[i * LD + 4 * q + k]; bv[q] = Bm[i * LD + 4 * q + k]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) a = mfma(-1e-6 * av[q], bv[q], a);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cc[(k + 4 * rr) * LD + i] = a[rr];
      }
    }
  } else if (MODE == BARRIER) {
    while (!__atomic_load_n(&stop, __ATOMIC_RELAXED)) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  if (keep == 123.456) sink[0] = keep;
}

template <int MODE, bool PRIO = false>
static void run(const double* dA, long long* dOut, double* dSink, int occ, int reps) {
  const int n_wg = 256 * occ;
  const size_t lds = (size_t)(160 * 1024 / occ) - 1024;    // exactly occ workgroups fit per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<MODE, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<long long> h(n_wg * 8);
  for (int pass = 0; pass < 2; ++pass) {       // pass 0 warms up
    hipMemset(dOut, 0, sizeof(long long) * n_wg * 8);
    hipLaunchKernelGGL((k_probe<MODE, PRIO>), dim3(n_wg), dim3(256), lds, 0, dA, dOut, reps, dSink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
  }
  hipMemcpy(h.data(), dOut, sizeof(long long) * n_wg * 8, hipMemcpyDeviceToHost);
  double ns = 0, cyc = 0, ns_max = 0; int cnt = 0;
  for (int b = 0; b < n_wg; ++b)
    for (int w = 0; w < (MODE == CHAIN ? 4 : 1); ++w) {
      const double t = 10.0 * h[(b * 4 + w) * 2] / reps, c = (double)h[(b * 4 + w) * 2 + 1] / reps;
      ns += t; cyc += c; ns_max = t > ns_max ? t : ns_max; ++cnt;
    }
  double mf = 0;
  if (MODE == MFMADEP || (MODE >= NOP2 && MODE <= NOP5)) {
    for (int b = 0; b < n_wg; ++b) mf += (double)h[(b * 4 + 1) * 2] / (10.0 * h[(b * 4) * 2]) / n_wg;   // MFMAs per ns of wave 1
    printf("      (filler wave: one MFMA every %.0f ns)\n", 1.0 / mf);
  }
  printf("  %d WG/CU  %s%-40s chain %7.0f ns (max %7.0f), %6.0f shader-clock ticks\n", occ, PRIO ? "[chain wave at s_setprio 3] " : "", kNames[MODE], ns / cnt, ns_max, cyc / cnt);
}

int main() {
  double hA[256];
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) hA[r * 16 + c] = (r == c ? 20.0 : 0.0) + 1.0 / (1.0 + abs(r - c));
  double *dA, *dSink; long long* dOut;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dSink, 64); hipMalloc(&dOut, sizeof(long long) * 256 * 3 * 8);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  const int reps = 200;
  for (int occ = 1; occ <= 3; ++occ) {
    run<IDLE>(dA, dOut, dSink, occ, reps);
    run<BARRIER>(dA, dOut, dSink, occ, reps);
    run<CHAIN>(dA, dOut, dSink, occ, reps);
    run<MFMA64>(dA, dOut, dSink, occ, reps);
    run<VALU64>(dA, dOut, dSink, occ, reps);
    run<VALU32>(dA, dOut, dSink, occ, reps);
    run<SCALAR>(dA, dOut, dSink, occ, reps);
    run<LDSRD>(dA, dOut, dSink, occ, reps);
    run<VMEM>(dA, dOut, dSink, occ, reps);
    run<MFMADEP>(dA, dOut, dSink, occ, reps);
    run<TRAIL>(dA, dOut, dSink, occ, reps);
    run<NOP2>(dA, dOut, dSink, occ, reps);
    run<NOP3>(dA, dOut, dSink, occ, reps);
    run<NOP4>(dA, dOut, dSink, occ, reps);
    run<NOP5>(dA, dOut, dSink, occ, reps);
    run<NOP4, true>(dA, dOut, dSink, occ, reps);
    run<MFMA64, true>(dA, dOut, dSink, occ, reps);
    run<MFMADEP, true>(dA, dOut, dSink, occ, reps);
    run<TRAIL, true>(dA, dOut, dSink, occ, reps);
    run<VALU64, true>(dA, dOut, dSink, occ, reps);
  }
  return 0;
}
