// EXPERIMENT, not part of the product (libcc_hip.so does not contain or call any of this).
//
// north_star speaks of a dense (6F+K) x (6F+K) Cholesky of the damped normal equations on the matrix cores. The product
// solves the same system exactly by block elimination; this file builds the dense route so that
// the two can be put side by side on MI355X: same matrix, same right-hand side, step compared entry by entry, time and
// matrix-core rate measured (scripts/dense_mfma/run_dense.py).
//
// Right-looking blocked Cholesky, 64 x 64 tiles, lower triangle of a row-major padded matrix in HBM:
//   per block column k:  potrf (one workgroup, tile in registers, also forms inv(L_kk))
//                        trsm  (one workgroup per tile below: A_ik <- A_ik inv(L_kk)^T, a 64x64x64 product on MFMA)
//                        syrk  (one workgroup per tile (i, j), k < j <= i: A_ij -= L_ik L_jk^T, same MFMA micro-kernel)
// The right-hand side rides along as one extra matrix row (row n of the padding), so the forward substitution is
// part of the factorisation; the backward substitution runs tile row by tile row.
// v_mfma_f64_16x16x4_f64: D[i][j] += sum_k A[i][k] B[k][j]; A operand: lane (i = lane & 15, k = lane >> 4);
// B operand: lane (j = lane & 15, k = lane >> 4); D: column lane & 15, row (lane >> 4) + 4 * reg.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace {

constexpr int NB = 64;     // tile edge
constexpr int LDT = 68;    // LDS row stride of a staged tile (doubles): 2-way bank conflicts at most on operand reads
typedef double d4 __attribute__((ext_vector_type(4)));

#define DN_HIP(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "dense_step: %s failed: %s\n", #x, hipGetErrorString(e_));      \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

// ---- diagonal tile: Cholesky + explicit inverse of the factor, one workgroup ------------------------------------
// Thread (row = t >> 2, q = t & 3) keeps A[row][q + 4 * it], it = 0..15, in registers for the whole factorisation; a
// column step broadcasts the current column through a double-buffered 64-entry LDS vector (one barrier per column):
// A[row][c] -= A[row][j] A[c][j] / A[j][j]. The upper triangle is carried along as garbage and never read.
__global__ __launch_bounds__(256) void k_potrf(double* A, int64_t ld, int k, double* W, int* fail) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  double (*sL)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(dyn_lds);
  double (*sW)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(dyn_lds + NB * (NB + 1));
  double (*col)[NB] = reinterpret_cast<double (*)[NB]>(dyn_lds + 2 * NB * (NB + 1));
  double* pivs = dyn_lds + 2 * NB * (NB + 1) + 2 * NB;
  const int tid = threadIdx.x, row = tid >> 2, q = tid & 3;
  double* T = A + ((int64_t)k * NB) * ld + (int64_t)k * NB;
  double a[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) a[it] = T[(int64_t)row * ld + q + 4 * it];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double* cb = col[j & 1];
    if (q == (j & 3)) cb[row] = a[j >> 2];
    __syncthreads();
    double piv = cb[j];
    if (!(piv > 0.0)) { bad = true; piv = 1.0; }
    const double cr = cb[row];
    if (q == (j & 3)) sL[row][j] = row >= j ? cr : 0.0;   // unscaled column; divided by sqrt(pivot) after the loop
    if (tid == 0) pivs[j] = piv;
    double ip = __builtin_amdgcn_rcp(piv);                // reciprocal by two Newton steps instead of a full division
    ip = ip * (2.0 - piv * ip);
    ip = ip * (2.0 - piv * ip);
    const double lr = cr * ip;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (4 * it + 3 <= j) continue;   // every column of this register lies at or left of the pivot
      const int c = q + 4 * it;
      if (c > j) a[it] -= lr * cb[c];
    }
  }
  if (bad && tid == 0) *fail = 1;
  __syncthreads();
  for (int e = tid; e < NB * NB; e += 256) { const int r = e >> 6, c = e & 63; sL[r][c] = sL[r][c] / sqrt(pivs[c]); }
  __syncthreads();
  // inverse of the lower-triangular factor: column c by forward substitution, the dot products split over 4 threads
  {
    const int c = tid >> 2, part = tid & 3;
    for (int i = 0; i < NB; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < 16; ++m) {   // static trip count: all LDS reads of a row are in flight together
        const int j = c + part + 4 * m;
        if (j < i) acc += sL[i][j] * sW[j][c];
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) sW[i][c] = i < c ? 0.0 : (i == c ? 1.0 / sL[c][c] : -acc / sL[i][i]);
      __syncthreads();
    }
  }
  double* Wk = W + (int64_t)k * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    T[(int64_t)r * ld + c] = sL[r][c];
    Wk[e] = sW[r][c];
  }
}

// ---- 64 x 64 x 64 product P Q^T on the matrix cores; P, Q row-major tiles staged in LDS --------------------------
__device__ __forceinline__ void load_tile(double (*dst)[LDT], const double* src, int64_t ld, int tid) {
  // 256 threads, 16 doubles each: thread t loads row t >> 2, columns 16 * (t & 3) .. + 15 (two 64-byte segments)
  const int r = tid >> 2, c0 = (tid & 3) * 16;
  const double2* p = reinterpret_cast<const double2*>(src + (int64_t)r * ld + c0);
#pragma unroll
  for (int q = 0; q < 8; ++q) { const double2 v = p[q]; dst[r][c0 + 2 * q] = v.x; dst[r][c0 + 2 * q + 1] = v.y; }
}

__device__ __forceinline__ void tile_pqt(double (*sP)[LDT], double (*sQ)[LDT], int wave, int lane, d4 acc[4]) {
  const int i = lane & 15, sub = lane >> 4;
#pragma unroll
  for (int ts = 0; ts < 16; ++ts) {
    const double a = sP[16 * wave + i][4 * ts + sub];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const double b = sQ[16 * ct + i][4 * ts + sub];
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ct], 0, 0, 0);
    }
  }
}

// A_ik <- A_ik inv(L_kk)^T for every tile row i > k
__global__ __launch_bounds__(256) void k_trsm(double* A, int64_t ld, int k, const double* W) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  double (*sP)[LDT] = reinterpret_cast<double (*)[LDT]>(dyn_lds);
  double (*sQ)[LDT] = reinterpret_cast<double (*)[LDT]>(dyn_lds + NB * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = k + 1 + blockIdx.x;
  double* T = A + ((int64_t)i * NB) * ld + (int64_t)k * NB;
  load_tile(sP, T, ld, tid);
  load_tile(sQ, W + (int64_t)k * NB * NB, NB, tid);
  __syncthreads();
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  tile_pqt(sP, sQ, wave, lane, acc);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(int64_t)(16 * wave + (lane >> 4) + 4 * r) * ld + 16 * ct + (lane & 15)] = acc[ct][r];
}

// A_ij -= L_ik L_jk^T for k < j <= i (2-D grid over the trailing tiles, upper ones exit)
__global__ __launch_bounds__(256) void k_syrk(double* A, int64_t ld, int k) {
  const int i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
  if (j > i) return;
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  double (*sP)[LDT] = reinterpret_cast<double (*)[LDT]>(dyn_lds);
  double (*sQ)[LDT] = reinterpret_cast<double (*)[LDT]>(dyn_lds + NB * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  load_tile(sP, A + ((int64_t)i * NB) * ld + (int64_t)k * NB, ld, tid);
  load_tile(sQ, A + ((int64_t)j * NB) * ld + (int64_t)k * NB, ld, tid);
  double* C = A + ((int64_t)i * NB) * ld + (int64_t)j * NB;
  d4 c0[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) c0[ct][r] = C[(int64_t)(16 * wave + (lane >> 4) + 4 * r) * ld + 16 * ct + (lane & 15)];
  __syncthreads();
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  tile_pqt(sP, sQ, wave, lane, acc);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(int64_t)(16 * wave + (lane >> 4) + 4 * r) * ld + 16 * ct + (lane & 15)] = c0[ct][r] - acc[ct][r];
}

// ---- backward substitution L^T d = y over the leading n unknowns ----------------------------------------------
__global__ void k_take_rhs(const double* A, int64_t ld, int n, int npad, double* r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) r[i] = i < n ? A[(int64_t)n * ld + i] : 0.0;   // row n of the factor = L^-1 b
}

// d_k = inv(L_kk)^T r_k
__global__ __launch_bounds__(256) void k_bdiag(const double* W, int k, const double* r, double* d) {
  __shared__ double sr[NB];
  __shared__ double part[4][NB];
  const int c = threadIdx.x & 63, p = threadIdx.x >> 6;
  if (threadIdx.x < NB) sr[threadIdx.x] = r[k * NB + threadIdx.x];
  __syncthreads();
  const double* Wk = W + (int64_t)k * NB * NB;
  double a = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int row = 16 * p + i; a += Wk[row * NB + c] * sr[row]; }   // W[row][c] = 0 for row < c
  part[p][c] = a;
  __syncthreads();
  if (p == 0) d[k * NB + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// r_j -= L_kj^T d_k for every j < k
__global__ __launch_bounds__(64) void k_bupdate(const double* A, int64_t ld, int k, const double* d, double* r) {
  __shared__ double sd[NB];
  const int c = threadIdx.x, j = blockIdx.x;
  sd[c] = d[k * NB + c];
  __syncthreads();
  const double* T = A + ((int64_t)k * NB) * ld + (int64_t)j * NB;
  double a = 0.0;
  for (int row = 0; row < NB; ++row) a += T[(int64_t)row * ld + c] * sd[row];
  r[j * NB + c] -= a;
}

}  // namespace

// Solves A d = b for the symmetric positive definite n x n matrix A (row-major, only the lower triangle is read)
// `reps` times on device `device`; returns the step and the average milliseconds of the factorisation (forward
// substitution included) and of the backward substitution, both measured with hipEvents on the stream.
extern "C" int dn_solve(int device, int n, const double* A_host, const double* b_host, double* d_host, int reps,
                        double* ms_factor, double* ms_back, double* mfma_flop) {
  DN_HIP(hipSetDevice(device));
  const int nt = (n + 1 + NB - 1) / NB;   // + 1: the right-hand-side row
  const int npad = nt * NB;
  const int64_t ld = npad;
  double *A0 = nullptr, *A = nullptr, *W = nullptr, *r = nullptr, *d = nullptr;
  int* fail = nullptr;
  DN_HIP(hipMalloc(&A0, sizeof(double) * ld * npad));
  DN_HIP(hipMalloc(&A, sizeof(double) * ld * npad));
  DN_HIP(hipMalloc(&W, sizeof(double) * (size_t)nt * NB * NB));
  DN_HIP(hipMalloc(&r, sizeof(double) * npad));
  DN_HIP(hipMalloc(&d, sizeof(double) * npad));
  DN_HIP(hipMalloc(&fail, sizeof(int)));
  DN_HIP(hipMemset(A0, 0, sizeof(double) * ld * npad));
  DN_HIP(hipMemset(fail, 0, sizeof(int)));
  DN_HIP(hipMemcpy2D(A0, sizeof(double) * ld, A_host, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice));
  {  // row n: the right-hand side, with a diagonal large enough to stay positive; identity on the rest of the padding
    std::vector<double> row((size_t)npad, 0.0);
    for (int i = 0; i < n; ++i) row[(size_t)i] = b_host[i];
    row[(size_t)n] = 1e30;
    DN_HIP(hipMemcpy(A0 + (int64_t)n * ld, row.data(), sizeof(double) * npad, hipMemcpyHostToDevice));
    for (int i = n + 1; i < npad; ++i) {
      const double one = 1.0;
      DN_HIP(hipMemcpy(A0 + (int64_t)i * ld + i, &one, sizeof(double), hipMemcpyHostToDevice));
    }
  }
  constexpr int kPotrfLds = (2 * NB * (NB + 1) + 3 * NB) * 8, kTileLds = 2 * NB * LDT * 8;
  DN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf), hipFuncAttributeMaxDynamicSharedMemorySize, kPotrfLds));
  DN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, kTileLds));
  DN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk), hipFuncAttributeMaxDynamicSharedMemorySize, kTileLds));
  hipStream_t st;
  DN_HIP(hipStreamCreate(&st));
  hipEvent_t e0, e1, e2;
  DN_HIP(hipEventCreate(&e0)); DN_HIP(hipEventCreate(&e1)); DN_HIP(hipEventCreate(&e2));
  double tf = 0.0, tb = 0.0;
  for (int rep = 0; rep < reps + 1; ++rep) {   // first pass warms up and is not timed
    DN_HIP(hipMemcpyAsync(A, A0, sizeof(double) * ld * npad, hipMemcpyDeviceToDevice, st));
    DN_HIP(hipEventRecord(e0, st));
    for (int k = 0; k < nt; ++k) {
      hipLaunchKernelGGL(k_potrf, dim3(1), dim3(256), kPotrfLds, st, A, ld, k, W, fail);
      const int m = nt - k - 1;
      if (m > 0) {
        hipLaunchKernelGGL(k_trsm, dim3(m), dim3(256), kTileLds, st, A, ld, k, W);
        hipLaunchKernelGGL(k_syrk, dim3(m, m), dim3(256), kTileLds, st, A, ld, k);
      }
    }
    DN_HIP(hipEventRecord(e1, st));
    hipLaunchKernelGGL(k_take_rhs, dim3((npad + 255) / 256), dim3(256), 0, st, A, ld, n, npad, r);
    for (int k = nt - 1; k >= 0; --k) {
      hipLaunchKernelGGL(k_bdiag, dim3(1), dim3(256), 0, st, W, k, r, d);
      if (k > 0) hipLaunchKernelGGL(k_bupdate, dim3(k), dim3(64), 0, st, A, ld, k, d, r);
    }
    DN_HIP(hipEventRecord(e2, st));
    DN_HIP(hipStreamSynchronize(st));
    DN_HIP(hipGetLastError());
    if (rep > 0) {
      float a = 0, b = 0;
      DN_HIP(hipEventElapsedTime(&a, e0, e1));
      DN_HIP(hipEventElapsedTime(&b, e1, e2));
      tf += a; tb += b;
    }
  }
  int hfail = 0;
  DN_HIP(hipMemcpy(&hfail, fail, sizeof(int), hipMemcpyDeviceToHost));
  DN_HIP(hipMemcpy(d_host, d, sizeof(double) * n, hipMemcpyDeviceToHost));
  *ms_factor = tf / reps;
  *ms_back = tb / reps;
  // flops issued to the matrix cores by one factorisation: every trsm / syrk tile product is 2 * 64^3
  double tiles = 0.0;
  for (int k = 0; k < nt; ++k) { const double m = nt - k - 1; tiles += m + m * (m + 1) / 2; }
  *mfma_flop = tiles * 2.0 * NB * NB * NB;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  (void)hipStreamDestroy(st);
  for (void* p : {(void*)A0, (void*)A, (void*)W, (void*)r, (void*)d, (void*)fail}) (void)hipFree(p);
  return hfail ? 2 : 0;
}
