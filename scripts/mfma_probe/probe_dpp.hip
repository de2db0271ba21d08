// Direction of DPP row_ror on gfx950: which lane does lane i read for row_ror:4 / row_ror:8?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_update_dpp(0, l, 0x124, 0xf, 0xf, false);        // row_ror:4
  out[64 + l] = __builtin_amdgcn_update_dpp(0, l, 0x128, 0xf, 0xf, false);   // row_ror:8
}
int main() {
  int h[128], *d;
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_ror:4  lane 0 reads lane %d, lane 5 reads %d, lane 17 reads %d, lane 63 reads %d\n", h[0], h[5], h[17], h[63]);
  printf("row_ror:8  lane 0 reads lane %d, lane 5 reads %d, lane 17 reads %d\n", h[64], h[69], h[81]);
  return 0;
}
