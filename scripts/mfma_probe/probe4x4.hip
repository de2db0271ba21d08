// Operand / result lane layout of v_mfma_f64_4x4x4_4b on gfx950, found by experiment: the lane index has
// three 2-bit fields; every assignment of (row/col, k, block) to the fields is tried for A, B and D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
  double A[4][4][4], B[4][4][4];
  for (int blk = 0; blk < 4; ++blk) for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) {
    A[blk][x][y] = 1.0 + 0.37 * blk + 1.3 * x + 0.11 * y * y;      // A[blk][i][k]
    B[blk][x][y] = 2.0 - 0.21 * blk + 0.7 * x * x + 1.9 * y;       // B[blk][k][j]
  }
  int found = 0;
  for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) {
    for (int l = 0; l < 64; ++l) {
      const int f[3] = {l & 3, (l >> 2) & 3, (l >> 4) & 3};
      // perms[p] = {field of row/col index, field of k, field of block}
      ha[l] = A[f[perms[pa][2]]][f[perms[pa][0]]][f[perms[pa][1]]];
      hb[l] = B[f[perms[pb][2]]][f[perms[pb][1]]][f[perms[pb][0]]];
    }
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    for (int po = 0; po < 6; ++po) {
      int ok = 1;
      for (int l = 0; l < 64 && ok; ++l) {
        const int f[3] = {l & 3, (l >> 2) & 3, (l >> 4) & 3};
        const int i = f[perms[po][0]], j = f[perms[po][1]], blk = f[perms[po][2]];
        double ref = 0;
        for (int kk = 0; kk < 4; ++kk) ref += A[blk][i][kk] * B[blk][kk][j];
        if (std::fabs(ref - hd[l]) > 1e-9) ok = 0;
      }
      if (ok) {
        ++found;
        printf("MATCH: A fields(i,k,blk)=(%d,%d,%d)  B fields(j,k,blk)=(%d,%d,%d)  D fields(i,j,blk)=(%d,%d,%d)   [field 0 = lane bits 0-1, 1 = bits 2-3, 2 = bits 4-5]\n",
               perms[pa][0], perms[pa][1], perms[pa][2], perms[pb][0], perms[pb][1], perms[pb][2], perms[po][0], perms[po][1], perms[po][2]);
      }
    }
  }
  printf("done, %d consistent layouts\n", found);
  return 0;
}
