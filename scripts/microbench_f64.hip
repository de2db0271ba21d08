// microbench_f64.hip -- measures on gfx950: (1) issue rate of v_mfma_f64_16x16x4_f64 and
// v_mfma_f64_4x4x4_4b_f64, (2) v_fma_f64 rate, (3) whether fp64 MFMA of one wave overlaps with fp64
// VALU of other waves on the same SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o mb scripts/microbench_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: mfma16 only, 1: valu only, 2: even waves mfma / odd waves valu, 3: mfma4x4 only, 4: both in every wave
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  const int wave = threadIdx.x >> 6;
  d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
  double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  double x = seed + threadIdx.x * 1e-3, y = seed * 0.5;
  double v0 = x, v1 = y, v2 = x + 1, v3 = y + 1, v4 = x + 2, v5 = y + 2, v6 = x + 3, v7 = y + 3;
  const bool do_mfma = MODE == 0 || MODE == 4 || (MODE == 2 && (wave & 1) == 0);
  const bool do_valu = MODE == 1 || MODE == 4 || (MODE == 2 && (wave & 1) == 1);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        b0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, b1, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, b2, 0, 0, 0);
        b3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, y, b3, 0, 0, 0);
      }
    }
    if (do_mfma) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
      }
    }
    if (do_valu) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        v0 = fma(v0, 1.0000001, 1e-9); v1 = fma(v1, 0.9999999, 1e-9); v2 = fma(v2, 1.0000001, 1e-9); v3 = fma(v3, 0.9999999, 1e-9);
        v4 = fma(v4, 1.0000001, 1e-9); v5 = fma(v5, 0.9999999, 1e-9); v6 = fma(v6, 1.0000001, 1e-9); v7 = fma(v7, 0.9999999, 1e-9);
      }
    }
  }
  double r = a0[0] + a1[1] + a2[2] + a3[3] + b0 + b1 + b2 + b3 + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(int blocks, int iters, double* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* d; hipMalloc(&d, 256 * 2048 * 8);
  const int iters = 20000;
  // one block of 4 waves per CU -> one wave per SIMD; 2 blocks per CU -> 2 waves per SIMD
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int blocks = 256 * bpc;
    const double t0 = run<0>(blocks, iters, d), t1 = run<1>(blocks, iters, d), t2 = run<2>(blocks, iters, d), t3 = run<3>(blocks, iters, d), t4 = run<4>(blocks, iters, d);
    const double mf = 16.0 * iters, vf = 128.0 * iters;  // instructions per wave
    const double waves_per_simd = bpc;
    printf("waves/SIMD=%d\n", bpc);
    printf("  mfma16x16x4 only : %.3f ms  -> %.1f ns per MFMA per SIMD, %.2f TFLOP/s chip\n", t0, t0 * 1e6 / (mf * waves_per_simd), 2048.0 * mf * 4 * blocks / (t0 * 1e-3) / 1e12);
    printf("  valu fma only    : %.3f ms  -> %.2f ns per v_fma_f64 per SIMD, %.2f TFLOP/s chip\n", t1, t1 * 1e6 / (vf * waves_per_simd), 128.0 * vf * 4 * blocks / (t1 * 1e-3) / 1e12);
    printf("  even mfma/odd valu: %.3f ms (sum of halves would be %.3f, max %.3f)\n", t2, 0.5 * (t0 + t1), 0.5 * (t0 > t1 ? t0 : t1));
    printf("  mfma4x4x4 only   : %.3f ms  -> %.1f ns per MFMA per SIMD, %.2f TFLOP/s chip\n", t3, t3 * 1e6 / (mf * waves_per_simd), 512.0 * mf * 4 * blocks / (t3 * 1e-3) / 1e12);
    printf("  both in each wave: %.3f ms (sum %.3f)\n", t4, t0 + t1);
  }
  return 0;
}
