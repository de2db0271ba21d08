// microbench_k2.hip -- issue rate of the accumulation pattern of k_rig_sweep_k2 on gfx950: acc[s] = fma(w[i], w[j], acc[s]) with all
// three operands in vector registers (66 accumulators, 14 row entries), against the same count of FMAs with constant operands.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/mbk2 scripts/microbench_k2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int WAVES>   // 0: three-register FMAs (pairs i >= j of 14 entries, 105 per row); 1: the same with a v_mul per entry in front; 2: constant-operand FMAs
__global__ __launch_bounds__(64, WAVES) void k(double* out, const double* in, int iters) {
  double acc[105];
#pragma unroll
  for (int e = 0; e < 105; ++e) acc[e] = 0.0;
  double w[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) w[i] = in[threadIdx.x + 64 * i];
  double s = in[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 14; ++i) w[i] = w[i] * s;
    }
    if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 105; ++e) acc[e] = fma(acc[e], 1.0000001, 1e-9);
    } else {
      int e = 0;
#pragma unroll
      for (int i = 0; i < 14; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) { acc[e] = fma(w[i], w[j], acc[e]); ++e; }
    }
    asm volatile("" ::: "memory");
  }
  double r = 0.0;
#pragma unroll
  for (int e = 0; e < 105; ++e) r += acc[e];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int MODE, int WAVES>
double run(int blocks, int iters, double* d, double* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64), 0, 0, d, in, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64), 0, 0, d, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *d, *in; hipMalloc(&d, 64 * 4096 * 8); hipMalloc(&in, 64 * 16 * 8);
  hipMemset(in, 0, 64 * 16 * 8);
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 1024 * wps;   // one-wave workgroups: wps waves per SIMD
    const double t0 = wps == 1 ? run<0, 1>(blocks, iters, d, in) : run<0, 2>(blocks, iters, d, in);
    const double t1 = wps == 1 ? run<1, 1>(blocks, iters, d, in) : run<1, 2>(blocks, iters, d, in);
    const double t2 = wps == 1 ? run<2, 1>(blocks, iters, d, in) : run<2, 2>(blocks, iters, d, in);
    const double n = 105.0 * iters * wps;
    printf("waves/SIMD=%d: three-register fma %.2f ns each (%.2f TFLOP/s chip), with 14 muls per row %.2f ns per instruction, constant-operand fma %.2f ns\n",
           wps, t0 * 1e6 / n, 128.0 * n * 1024 / (t0 * 1e-3) / 1e12 / wps * wps, t1 * 1e6 / (119.0 * iters * wps), t2 * 1e6 / n);
  }
  return 0;
}
