// Feasibility probe for the mailbox exchange: two processes, IPC-mapped uncached device memory, a
// kernel in A spinning (bounded) on a flag that a kernel in B sets after writing data.
//   probe A <file>   : allocate, export handle to <file>, wait for B's post, print what arrived
//   probe B <file>   : import, post
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

__global__ void k_wait(double* box, double* out, long long limit_ticks) {
  unsigned long long* flag = reinterpret_cast<unsigned long long*>(box + 127);
  const long long t0 = wall_clock64();
  int ok = 1;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 7ull) {
    if (wall_clock64() - t0 > limit_ticks) { ok = 0; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  out[0] = ok;
  out[1] = box[0];
  out[2] = box[111];
  out[3] = (double)(wall_clock64() - t0);
}

__global__ void k_post(double* box) {
  const int t = threadIdx.x;
  if (t < 112) __builtin_nontemporal_store(1000.0 + t, box + t);
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(box + 127), 7ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  const bool A = argv[1][0] == 'A';
  const int flags = argc > 3 ? atoi(argv[3]) : 3;  // 3 = uncached, 1 = fine-grained, 0 = plain hipMalloc
  CK(hipSetDevice(0));
  if (A) {
    double* box = nullptr;
    if (flags == 0) CK(hipMalloc(&box, 4096));
    else CK(hipExtMallocWithFlags((void**)&box, 4096, flags == 3 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    CK(hipMemset(box, 0, 4096));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hnd;
    CK(hipIpcGetMemHandle(&hnd, box));
    FILE* f = fopen(argv[2], "wb");
    fwrite(&hnd, sizeof(hnd), 1, f);
    fclose(f);
    double* out;
    CK(hipHostMalloc(&out, 64));
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, 0, box, out, 100000000LL * 8);  // 8 s at 100 MHz
    CK(hipDeviceSynchronize());
    printf("A: ok=%g data0=%g data111=%g waited_ms=%.3f (flags=%d)\n", out[0], out[1], out[2], out[3] / 1e5, flags);
    return out[0] == 1.0 && out[1] == 1000.0 && out[2] == 1111.0 ? 0 : 3;
  } else {
    hipIpcMemHandle_t hnd;
    for (int i = 0; i < 100; ++i) { if (access(argv[2], R_OK) == 0) break; usleep(100000); }
    usleep(200000);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(&hnd, sizeof(hnd), 1, f) != 1) { printf("B: no handle\n"); return 2; }
    fclose(f);
    double* box = nullptr;
    CK(hipIpcOpenMemHandle((void**)&box, hnd, hipIpcMemLazyEnablePeerAccess));
    hipLaunchKernelGGL(k_post, dim3(1), dim3(128), 0, 0, box);
    CK(hipDeviceSynchronize());
    printf("B: posted\n");
    CK(hipIpcCloseMemHandle(box));
    return 0;
  }
}
