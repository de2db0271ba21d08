// Probe for the seam the round-3 review proposed for the intrinsics persistent kernel: G worker workgroups add their 80
// elimination sums as fixed-point int64 ATOMICS into one set of 80 accumulators and arrive on a counter; a control workgroup
// polls the counter and reads the 80 sums in one round trip -- instead of the two gather hops (rows -> leaders -> control,
// 2.2 + 2.1 us in profiles/r03/intr_persist_marks.jsonl). Measured here on its own: the time from the moment the LAST
// worker has its values ready to the moment the control holds the sums, over many rounds of one resident launch.
//   hipcc -O3 --offload-arch=gfx950 probe.hip -o probe && ./probe [G] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline long long wall() { return __builtin_readcyclecounter() * 0 + (long long)wall_clock64(); }

// mode 0: atomics + counter; mode 1: self-validating rows (the product's seam: {tag : half} words), leaders of 16, control
struct Args {
  unsigned long long* acc;      // [2][8][80] fixed-point accumulators (parity by round; mode 2: one set per XCD, mode 0: set 0 only)
  unsigned* cnt;                // [2][8] arrive counters
  unsigned long long* go;       // [1] control -> workers: round number (self-validating word)
  unsigned long long* rows;     // [G][160] rows (mode 1)
  unsigned long long* lrows;    // [G/16][160] leader rows (mode 1)
  long long* t_ready;           // [rounds][G] wall clock when worker g had its values
  long long* t_done;            // [rounds] wall clock when the control had the sums
  double* out;                  // [rounds] checksum
  unsigned* fail;
  int G, rounds, mode;
};

__device__ inline bool gave_up(long long t0, unsigned* fail) {   // 0.2 s of the 100 MHz clock, or somebody else gave up
  if (wall_clock64() - t0 > 20000000LL) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
  return __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
__device__ inline unsigned long long ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_probe(Args a) {
  const int tid = threadIdx.x, g = blockIdx.x;
  __shared__ double s_sum[80];
  if (g == a.G) {   // ---- control
    for (int r = 0; r < a.rounds; ++r) {
      const unsigned tag = r + 1;
      if (tid == 0) st(a.go, tag);
      double v = 0.0;
      if (a.mode == 2) {
        // one accumulator set and one counter per XCD: a word is hit by the ~G / 8 workers of one XCD, the control reads 8 x 80
        // words and 8 counters in one round trip each
        if (tid < 8) {
          const long long t0 = wall_clock64();
          for (;;) {
            const unsigned c = __hip_atomic_load(a.cnt + ((r & 1) * 8 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned tot = c;
            for (int o = 1; o < 8; o <<= 1) tot += __shfl_xor(tot, o, 8);
            if (tot == (unsigned)a.G || gave_up(t0, a.fail)) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        __syncthreads();
        if (tid < 80) {
          long long f = 0;
          unsigned long long w[8];
          for (int x = 0; x < 8; ++x) w[x] = ld(a.acc + ((r & 1) * 8 + x) * 80 + tid);
          for (int x = 0; x < 8; ++x) f += (long long)w[x];
          v = (double)f * (1.0 / 1099511627776.0);
        }
        __syncthreads();
        for (int i = tid; i < 8 * 80; i += 256) st(a.acc + (r & 1) * 8 * 80 + i, 0ull);
        if (tid < 8) __hip_atomic_store(a.cnt + (r & 1) * 8 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (a.mode == 0) {
        if (tid == 0) { const long long t0 = wall_clock64(); while (__hip_atomic_load(a.cnt + (r & 1) * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)a.G && !gave_up(t0, a.fail)) __builtin_amdgcn_s_sleep(1); }
        __syncthreads();
        if (tid < 80) {
          const long long f = (long long)ld(a.acc + (r & 1) * 8 * 80 + tid);
          v = (double)f * (1.0 / 1099511627776.0);   // 2^-40
        }
        __syncthreads();
        // reset this parity for round r + 2 (nobody touches it before the next-but-one round's go)
        if (tid < 80) st(a.acc + (r & 1) * 8 * 80 + tid, 0ull);
        if (tid == 0) __hip_atomic_store(a.cnt + (r & 1) * 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const int nl = (a.G + 15) / 16;
        if (tid < 160) {
          // column tid >> 1, half tid & 1 of every leader row
          unsigned long long w[16];
          const long long t0 = wall_clock64();
          for (;;) {
            bool ok = true;
            for (int l = 0; l < 16; ++l) { w[l] = ld(a.lrows + (size_t)(l < nl ? l : 0) * 160 + tid); ok = ok && (unsigned)(w[l] >> 32) == tag; }
            if (ok || gave_up(t0, a.fail)) break;
            __builtin_amdgcn_s_sleep(1);
          }
          // (pairs of threads hold the halves; reassemble through LDS)
          __shared__ unsigned s_h[16][160];
          for (int l = 0; l < 16; ++l) s_h[l][tid] = (unsigned)w[l];
          __syncthreads();
          if ((tid & 1) == 0) {
            for (int l = 0; l < nl; ++l) v += __longlong_as_double((long long)(((unsigned long long)s_h[l][tid + 1] << 32) | s_h[l][tid]));
          }
        } else __syncthreads();
      }
      if (tid < 160) s_sum[tid >> 1] = 0.0;
      __syncthreads();
      if (a.mode != 1 ? tid < 80 : (tid < 160 && (tid & 1) == 0)) s_sum[a.mode != 1 ? tid : tid >> 1] = v;
      __syncthreads();
      if (tid == 0) {
        a.t_done[r] = wall();
        double c = 0.0;
        for (int k = 0; k < 80; ++k) c += s_sum[k];
        a.out[r] = c;
      }
      __syncthreads();
    }
    return;
  }
  // ---- worker
  for (int r = 0; r < a.rounds; ++r) {
    const unsigned tag = r + 1;
    if (tid == 0) { const long long t0 = wall_clock64(); while ((unsigned)ld(a.go) != tag && !gave_up(t0, a.fail)) __builtin_amdgcn_s_sleep(1); }
    __syncthreads();
    // "work": a few hundred cycles that differ per workgroup, then the row is ready
    double v = 1.0 + 1e-3 * (tid % 80) + 1e-6 * g;
    for (int k = 0; k < 50 + (g & 7) * 10; ++k) v = v * 1.0000001 + 1e-9;
    if (tid == 0) a.t_ready[(size_t)r * a.G + g] = wall();
    if (a.mode == 0 || a.mode == 2) {
      unsigned xcc = 0;
      if (a.mode == 2) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
      xcc &= 7u;
      if (tid < 80) {
        const long long f = (long long)(v * 1099511627776.0);
        __hip_atomic_fetch_add(reinterpret_cast<long long*>(a.acc) + ((r & 1) * 8 + xcc) * 80 + tid, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.cnt + (r & 1) * 8 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tid < 160) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(tid < 160 ? (1.0 + 1e-3 * ((tid >> 1) % 80) + 1e-6 * g) : 0.0);
        (void)v;
        st(a.rows + (size_t)g * 160 + tid, ((unsigned long long)tag << 32) | ((tid & 1) ? (bits >> 32) : (bits & 0xffffffffull)));
      }
      if ((g & 15) == 0 && tid < 160) {
        const int n = a.G - g < 16 ? a.G - g : 16;
        unsigned long long w[16];
        const long long t0 = wall_clock64();
        for (;;) {
          bool ok = true;
          for (int l = 0; l < 16; ++l) { w[l] = ld(a.rows + (size_t)(g + (l < n ? l : 0)) * 160 + tid); ok = ok && (unsigned)(w[l] >> 32) == tag; }
          if (ok || gave_up(t0, a.fail)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        __shared__ unsigned s_h[16][160];
        for (int l = 0; l < 16; ++l) s_h[l][tid] = (unsigned)w[l];
        __syncthreads();
        double sum = 0.0;
        const int c0 = tid & ~1;
        for (int l = 0; l < n; ++l) sum += __longlong_as_double((long long)(((unsigned long long)s_h[l][c0 + 1] << 32) | s_h[l][c0]));
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sum);
        st(a.lrows + (size_t)(g >> 4) * 160 + tid, ((unsigned long long)tag << 32) | ((tid & 1) ? (bits >> 32) : (bits & 0xffffffffull)));
      } else if ((g & 15) == 0) __syncthreads();
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 250, rounds = argc > 2 ? atoi(argv[2]) : 200;
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 1) continue;   // (the product's own seam is measured in the product: profiles/r03/intr_persist_marks.jsonl, 2.2 + 2.1 us)
    Args a{};
    a.G = G; a.rounds = rounds; a.mode = mode;
    CK(hipMalloc(&a.acc, 2 * 8 * 80 * 8)); CK(hipMemset(a.acc, 0, 2 * 8 * 80 * 8));
    CK(hipMalloc(&a.cnt, 64)); CK(hipMemset(a.cnt, 0, 64));
    CK(hipMalloc(&a.go, 8)); CK(hipMemset(a.go, 0, 8));
    CK(hipMalloc(&a.fail, 8)); CK(hipMemset(a.fail, 0, 8));
    CK(hipMalloc(&a.rows, (size_t)G * 160 * 8)); CK(hipMemset(a.rows, 0, (size_t)G * 160 * 8));
    CK(hipMalloc(&a.lrows, (size_t)((G + 15) / 16) * 160 * 8)); CK(hipMemset(a.lrows, 0, (size_t)((G + 15) / 16) * 160 * 8));
    CK(hipMalloc(&a.t_ready, (size_t)rounds * G * 8));
    CK(hipMalloc(&a.t_done, (size_t)rounds * 8));
    CK(hipMalloc(&a.out, (size_t)rounds * 8));
    hipLaunchKernelGGL(k_probe, dim3(G + 1), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<long long> tr((size_t)rounds * G), td(rounds);
    std::vector<double> out(rounds);
    CK(hipMemcpy(tr.data(), a.t_ready, tr.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(td.data(), a.t_done, td.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data(), a.out, out.size() * 8, hipMemcpyDeviceToHost));
    unsigned failed = 0;
    CK(hipMemcpy(&failed, a.fail, 4, hipMemcpyDeviceToHost));
    if (failed) { printf("{\"seam\": %d, \"error\": \"a wait gave up after 0.2 s\"}\n", mode); fflush(stdout); continue; }
    std::vector<double> lat;
    for (int r = 10; r < rounds; ++r) {
      long long last = 0;
      for (int g = 0; g < G; ++g) last = std::max(last, tr[(size_t)r * G + g]);
      lat.push_back((td[r] - last) / 100.0);
    }
    std::sort(lat.begin(), lat.end());
    printf("{\"seam\": \"%s\", \"workers\": %d, \"rounds\": %d, \"us_from_last_worker_ready_to_control_has_sums\": {\"median\": %.2f, \"p10\": %.2f, \"p90\": %.2f}, \"checksum\": %.9f}\n",
           mode == 0 ? "int64 fixed-point atomics into 80 accumulators + arrive counter" : "int64 fixed-point atomics, one accumulator set and counter per XCD (8 x 80 words)", G, rounds,
           lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10], out[rounds - 1]);
    fflush(stdout);
    hipFree(a.acc); hipFree(a.cnt); hipFree(a.go); hipFree(a.rows); hipFree(a.lrows); hipFree(a.t_ready); hipFree(a.t_done); hipFree(a.out);
  }
  return 0;
}
