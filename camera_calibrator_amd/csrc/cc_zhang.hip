// cc_zhang.hip -- Zhang's closed-form initialisation on the device: the step right before the LM
// hot path in Calibrator::Estimate (/root/reference/src/calibrator.cpp:47-66, geometry.cpp:70-203).
//
//   k_zhang_gram   : per frame, rows of the DLT system (geometry.cpp:79-97) -> A^T A (9x9, fp64) with
//                    the same LDS-staged v_mfma_f64_16x16x4_f64 contraction as the Jacobian sweep.
//   k_zhang_eig9   : one thread per frame: homography = eigenvector of the smallest eigenvalue of A^T A (= last
//                    right singular vector of A, which the reference takes from Eigen::JacobiSVD) by inverse
//                    iteration in registers, stored as float.
//   k_zhang_k      : one block: Zhang's V b = 0 system (geometry.cpp:123-177) as 6x6 normal equations,
//                    smallest eigenvector by inverse iteration, closed-form K.
//   k_zhang_poses  : one thread per frame: RecoverExtrinsics + FixRotationMatrix (geometry.cpp:179-203)
//                    and the quaternion of calibrator.cpp:63.
// The normal-equation form squares the condition number of the DLT system; on the generator's data
// the homography agrees with the SVD route to ~1e-7 relative, i.e. float32 resolution, which is the
// precision the reference keeps H in (H.cast<float>(), geometry.cpp:104).
#include <vector>

#include "cc_common.hpp"
#include "cc_device.hpp"

namespace cc {

constexpr int kZhangThreads = 256;
constexpr int kZhangLdsBytes = (4 * kStageDoublesPerWave) * 8;

// Cyclic Jacobi eigen-decomposition of a symmetric N x N matrix held in registers. On return the
// diagonal of A holds the eigenvalues and the columns of V the eigenvectors.
template <int N>
__device__ __forceinline__ void jacobi_eigen(double (&A)[N][N], double (&V)[N][N], int sweeps) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sw = 0; sw < sweeps; ++sw) {
#pragma unroll
    for (int p = 0; p < N - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        const double apq = A[p][q];
        const bool skip = apq == 0.0;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * (skip ? 1.0 : apq));
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (skip) t = 0.0;
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        A[p][p] -= t * apq;
        A[q][q] += t * apq;
        A[p][q] = 0.0;
        A[q][p] = 0.0;
#pragma unroll
        for (int r = 0; r < N; ++r) {
          if (r != p && r != q) {
            const double arp = A[r][p], arq = A[r][q];
            const double np = c * arp - s * arq, nq = s * arp + c * arq;
            A[r][p] = np; A[p][r] = np;
            A[r][q] = nq; A[q][r] = nq;
          }
          const double vrp = V[r][p], vrq = V[r][q];
          V[r][p] = c * vrp - s * vrq;
          V[r][q] = s * vrp + c * vrq;
        }
      }
  }
}

// Unit eigenvector of the SMALLEST eigenvalue of a symmetric positive semi-definite N x N matrix (registers):
// inverse iteration x <- A^-1 x with A^-1 applied as S (S A S + mu I)^-1 S, S = diag(A)^-1/2. The equilibration
// matters: DLT Gram matrices are badly graded (no Hartley normalisation in the reference, geometry.cpp:70-105; a
// nearly edge-on board gives world coordinates of 1e5 and eigenvalues from 1e-1 to 1e16), and a Cholesky of the raw
// matrix loses the small eigen-space entirely there, while the unit-diagonal one keeps it to ~1e-6 (float32, the
// precision H is stored in). mu = 1e-14 N keeps the pivots positive when the smallest eigenvalue is zero up to
// rounding (exact data, four-point homographies). The smallest eigenvalue is noise-sized and the next one is not,
// so the iteration gains many digits per step; 12 steps, ~1.5 kflop in ~70 registers, against ~26 kflop and 162
// live doubles (spilling) for the full Jacobi decomposition it replaces.
template <int N>
__device__ __forceinline__ void smallest_eigvec(const double (&A)[N][N], double (&x)[N]) {
  double sc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) sc[i] = A[i][i] > 0.0 ? rsqrt(A[i][i]) : 1.0;
  const double mu = 1e-14 * N;
  double L[N][N], inv[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = sc[j] * A[j][j] * sc[j] + mu;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    d = fmax(d, 1e-30);
    const double r = rsqrt(d);
    inv[j] = r;
    L[j][j] = d * r;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double a = sc[i] * A[i][j] * sc[j];
#pragma unroll
      for (int k = 0; k < j; ++k) a -= L[i][k] * L[j][k];
      L[i][j] = a * r;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (i & 1) ? 0.7 : 1.0;   // not orthogonal to anything in particular
  for (int it = 0; it < 12; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double a = x[i] * sc[i];
#pragma unroll
      for (int k = 0; k < i; ++k) a -= L[i][k] * x[k];
      x[i] = a * inv[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      double a = x[i];
#pragma unroll
      for (int k = i + 1; k < N; ++k) a -= L[k][i] * x[k];
      x[i] = a * inv[i];
    }
    double n2 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] *= sc[i]; n2 += x[i] * x[i]; }
    const double rn = rsqrt(n2);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] *= rn;
  }
}

// ---- per-frame Gram of the DLT rows ------------------------------------------------------------
__global__ __launch_bounds__(kZhangThreads, 4) void k_zhang_gram(int64_t F, const int64_t* off, const float* uv,
                                                                 const float* xyz, double* gram /*[F][256]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_stage = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t f = blockIdx.x;
  const int64_t s0 = off[f], s1 = off[f + 1];
  const int npass = (int)((s1 - s0 + kZhangThreads - 1) / kZhangThreads);
  double* stage = s_stage + wave * kStageDoublesPerWave;
  const float2* uv2 = reinterpret_cast<const float2*>(uv);
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int p = 0; p < npass; ++p) {
    const int64_t idx = s0 + (int64_t)p * kZhangThreads + tid;
    const bool valid = idx < s1;
    const int64_t ic = valid ? idx : s0;
    // p1 = world point (x, y only), p2 = image point; products in float as geometry.cpp:86-96
    const float x1 = xyz[ic * 3], y1 = xyz[ic * 3 + 1];
    const float2 m = uv2[ic];
    const float x2 = m.x, y2 = m.y;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
    if (valid) {
      v[3] = -x1; v[4] = -y1; v[5] = -1.0;
      v[6] = x1 * y2; v[7] = y1 * y2; v[8] = y2;
    }
    stage_row(stage, lane, v);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
    if (valid) {
      v[0] = x1; v[1] = y1; v[2] = 1.0;
      v[6] = -x1 * x2; v[7] = -y1 * x2; v[8] = -x2;
    }
    stage_row(stage, lane, v);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
  }
  __syncthreads();
  double* s_blk = s_stage;
#pragma unroll
  for (int r = 0; r < 4; ++r) s_blk[wave * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
  __syncthreads();
  gram[f * 256 + tid] = (s_blk[tid] + s_blk[256 + tid]) + (s_blk[512 + tid] + s_blk[768 + tid]);
}

// ---- homography = eigenvector of the smallest eigenvalue of A^T A ----------------------------------
__global__ __launch_bounds__(64) void k_zhang_eig9(int64_t F, const double* gram, float* Hs /*[F][9] row-major*/) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double A[9][9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) A[i][j] = gram[f * 256 + i * 16 + j];
  double h[9];
  smallest_eigvec<9>(A, h);
#pragma unroll
  for (int i = 0; i < 9; ++i) Hs[f * 9 + i] = (float)h[i];
}

// ---- K from the homographies (geometry.cpp:123-177) -------------------------------------------------
__device__ __forceinline__ void zhang_vij(const double* H, int i, int j, double* v) {
  // H row-major; H.col(i)(r) = H[r*3+i]
  v[0] = H[0 + i] * H[0 + j];
  v[1] = H[0 + i] * H[3 + j] + H[3 + i] * H[0 + j];
  v[2] = H[3 + i] * H[3 + j];
  v[3] = H[6 + i] * H[0 + j] + H[0 + i] * H[6 + j];
  v[4] = H[6 + i] * H[3 + j] + H[3 + i] * H[6 + j];
  v[5] = H[6 + i] * H[6 + j];
}

__global__ __launch_bounds__(256) void k_zhang_k(int64_t F, const float* Hs, float* K9) {
  __shared__ double s_w[4][24];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double g[21];  // upper triangle of the 6x6 normal matrix
#pragma unroll
  for (int e = 0; e < 21; ++e) g[e] = 0.0;
  for (int64_t f = tid; f < F; f += 256) {
    double H[9], a[6], b0[6], b1[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = (double)Hs[f * 9 + k];
    zhang_vij(H, 0, 1, a);
    zhang_vij(H, 0, 0, b0);
    zhang_vij(H, 1, 1, b1);
    double r2[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r2[k] = b0[k] - b1[k];
    int e = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) { g[e] += a[i] * a[j] + r2[i] * r2[j]; ++e; }
  }
#pragma unroll
  for (int e = 0; e < 21; ++e) {
    const double s = wave_sum(g[e]);
    if (lane == 0) s_w[wave][e] = s;
  }
  __syncthreads();
  if (tid != 0) return;
  double A[6][6];
  {
    int e = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) {
        const double s = (s_w[0][e] + s_w[1][e]) + (s_w[2][e] + s_w[3][e]);
        A[i][j] = s; A[j][i] = s;
        ++e;
      }
  }
  A[1][1] += (double)F * (double)F;  // zero-skew row (0, n, 0, 0, 0, 0), geometry.cpp:150-152
  double b[6];
  smallest_eigvec<6>(A, b);
  const double B11 = b[0], B12 = b[1], B22 = b[2], B13 = b[3], B23 = b[4], B33 = b[5];
  const double den = B11 * B22 - B12 * B12;
  const double v0 = (B12 * B13 - B11 * B23) / den;
  const double lambda = B33 - (B13 * B13 + v0 * (B12 * B13 - B11 * B23)) / B11;
  const double alpha = sqrt(lambda / B11);
  const double beta = sqrt(lambda * B11 / den);
  const double u0 = -B13 * alpha * alpha / lambda;  // skew forced to zero (geometry.cpp:168-171)
  K9[0] = (float)alpha; K9[1] = 0.0f; K9[2] = (float)u0;
  K9[3] = 0.0f; K9[4] = (float)beta; K9[5] = (float)v0;
  K9[6] = 0.0f; K9[7] = 0.0f; K9[8] = 1.0f;
}

// ---- poses ---------------------------------------------------------------------------------------
__device__ __forceinline__ void inv3f_dev(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float id = 1.0f / (m[0] * c00 + m[1] * c01 + m[2] * c02);
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__global__ __launch_bounds__(64) void k_zhang_poses(int64_t F, const float* Hs, const float* K9, float* q_out, float* t_out) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  float K[9], Ki[9], H[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { K[i] = K9[i]; H[i] = Hs[f * 9 + i]; }
  inv3f_dev(K, Ki);
  double Rp[9];
  float t[3];
  for (int attempt = 0; attempt < 2; ++attempt) {
    // RecoverExtrinsics, float arithmetic like the reference (geometry.cpp:179-195)
    float a0[3], r0[3], r1[3], r2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) a0[r] = Ki[r * 3] * H[0] + Ki[r * 3 + 1] * H[3] + Ki[r * 3 + 2] * H[6];
    const float l = 1.0f / sqrtf(a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]);
    float lK[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) lK[i] = l * Ki[i];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      r0[r] = l * a0[r];
      r1[r] = lK[r * 3] * H[1] + lK[r * 3 + 1] * H[4] + lK[r * 3 + 2] * H[7];
      t[r] = lK[r * 3] * H[2] + lK[r * 3 + 1] * H[5] + lK[r * 3 + 2] * H[8];
    }
    r2[0] = r0[1] * r1[2] - r0[2] * r1[1]; r2[1] = r0[2] * r1[0] - r0[0] * r1[2]; r2[2] = r0[0] * r1[1] - r0[1] * r1[0];
    // FixRotationMatrix: U V^T of the SVD = R (R^T R)^-1/2, via the eigen-decomposition of R^T R
    double R[3][3] = {{r0[0], r1[0], r2[0]}, {r0[1], r1[1], r2[1]}, {r0[2], r1[2], r2[2]}};
    double S[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) S[i][j] = R[0][i] * R[0][j] + R[1][i] * R[1][j] + R[2][i] * R[2][j];
    jacobi_eigen<3>(S, V, 10);
    double M[3][3];  // (R^T R)^-1/2 = V diag(1/sqrt(lambda)) V^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        M[i][j] = V[i][0] * V[j][0] / sqrt(S[0][0]) + V[i][1] * V[j][1] / sqrt(S[1][1]) + V[i][2] * V[j][2] / sqrt(S[2][2]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rp[i * 3 + j] = (double)(float)(R[i][0] * M[0][j] + R[i][1] * M[1][j] + R[i][2] * M[2][j]);
    if (t[2] >= 0.0f) break;
    // -H is the same homography with the board in front of the camera (see Calibrator::Estimate)
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = -H[i];
  }
  // Quaternionf(R): Shepperd's method as Eigen does it
  double q[4];
  double tr = Rp[0] + Rp[4] + Rp[8];
  if (tr > 0) {
    tr = sqrt(tr + 1.0);
    q[0] = 0.5 * tr;
    tr = 0.5 / tr;
    q[1] = (Rp[7] - Rp[5]) * tr; q[2] = (Rp[2] - Rp[6]) * tr; q[3] = (Rp[3] - Rp[1]) * tr;
  } else {
    int i = 0;
    if (Rp[4] > Rp[0]) i = 1;
    if (Rp[8] > Rp[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    tr = sqrt(Rp[i * 4] - Rp[j * 4] - Rp[k * 4] + 1.0);
    q[1 + i] = 0.5 * tr;
    tr = 0.5 / tr;
    q[0] = (Rp[k * 3 + j] - Rp[j * 3 + k]) * tr;
    q[1 + j] = (Rp[j * 3 + i] + Rp[i * 3 + j]) * tr;
    q[1 + k] = (Rp[k * 3 + i] + Rp[i * 3 + k]) * tr;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) q_out[f * 4 + i] = (float)q[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t_out[f * 3 + i] = t[i];
}

}  // namespace cc

namespace cc {
// The four kernels on arrays that are already on the device (cc_intrinsics_estimate: the handle's own copies), on
// `stream`. Scratch: gram double[F][256], H float[9F]; outputs K float[9], q float[4F], t float[3F] (device).
int zhang_on_device(hipStream_t stream, int64_t F, const int64_t* doff, const float* duv, const float* dxyz,
                    double* dgram, float* dH, float* dK, float* dq, float* dt) {
  hipLaunchKernelGGL(k_zhang_gram, dim3((unsigned)F), dim3(kZhangThreads), kZhangLdsBytes, stream, F, doff, duv, dxyz, dgram);
  hipLaunchKernelGGL(k_zhang_eig9, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, stream, F, dgram, dH);
  hipLaunchKernelGGL(k_zhang_k, dim3(1), dim3(256), 0, stream, F, dH, dK);
  hipLaunchKernelGGL(k_zhang_poses, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, stream, F, dH, dK, dq, dt);
  CC_HIP(hipGetLastError());
  return CC_OK;
}
}  // namespace cc

extern "C" int cc_zhang_init(int32_t device, int64_t F, const int64_t* off, const float* uv, const float* xyz,
                             float* K9, float* q_wxyz, float* t_xyz, float* homographies) {
  using namespace cc;
  if (F < 3 || !off || !K9) return fail(CC_ERR_BAD_ARGUMENT, "cc_zhang_init: needs >= 3 frames");
  if (off[0] != 0) return fail(CC_ERR_BAD_ARGUMENT, "frame_offsets[0] must be 0");
  for (int64_t f = 0; f < F; ++f)
    if (off[f + 1] - off[f] < 4) return fail(CC_ERR_BAD_ARGUMENT, "cc_zhang_init: frame %lld has fewer than 4 points", (long long)f);
  const int64_t N = off[F];
  if (F >= ((int64_t)1 << 28) || N >= ((int64_t)1 << 40))   // launch grids are 32-bit
    return fail(CC_ERR_BAD_ARGUMENT, "cc_zhang_init: problem too large (frames < 2^28, observations < 2^40)");
  if (!uv || !xyz) return fail(CC_ERR_BAD_ARGUMENT, "cc_zhang_init: NULL arrays");
  if (int rc = select_device(device)) return rc;
  // one scratch arena (one hipMalloc / hipFree per call), 256-byte aligned pieces
  size_t cursor = 0;
  auto take = [&](size_t bytes) { const size_t at = cursor; cursor += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_uv = take((size_t)N * 2 * sizeof(float)), o_xyz = take((size_t)N * 3 * sizeof(float));
  const size_t o_off = take((size_t)(F + 1) * sizeof(int64_t)), o_gram = take((size_t)F * 256 * sizeof(double));
  const size_t o_H = take((size_t)F * 9 * sizeof(float)), o_K = take(9 * sizeof(float));
  const size_t o_q = take((size_t)F * 4 * sizeof(float)), o_t = take((size_t)F * 3 * sizeof(float));
  char* arena = nullptr;
  struct Release {  // frees the scratch arena on every exit path
    char** p;
    ~Release() { if (*p) hipFree(*p); }
  } release{&arena};
  CC_HIP(hipMalloc(&arena, cursor));
  float* duv = reinterpret_cast<float*>(arena + o_uv);
  float* dxyz = reinterpret_cast<float*>(arena + o_xyz);
  int64_t* doff = reinterpret_cast<int64_t*>(arena + o_off);
  double* dgram = reinterpret_cast<double*>(arena + o_gram);
  float* dH = reinterpret_cast<float*>(arena + o_H);
  float* dK = reinterpret_cast<float*>(arena + o_K);
  float* dq = reinterpret_cast<float*>(arena + o_q);
  float* dt = reinterpret_cast<float*>(arena + o_t);
  CC_HIP(hipMemcpy(duv, uv, (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice));
  CC_HIP(hipMemcpy(dxyz, xyz, (size_t)N * 3 * sizeof(float), hipMemcpyHostToDevice));
  CC_HIP(hipMemcpy(doff, off, (size_t)(F + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (int rc = zhang_on_device(nullptr, F, doff, duv, dxyz, dgram, dH, dK, dq, dt)) return rc;
  CC_HIP(hipMemcpy(K9, dK, 9 * sizeof(float), hipMemcpyDeviceToHost));
  if (q_wxyz) CC_HIP(hipMemcpy(q_wxyz, dq, (size_t)F * 4 * sizeof(float), hipMemcpyDeviceToHost));
  if (t_xyz) CC_HIP(hipMemcpy(t_xyz, dt, (size_t)F * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (homographies) CC_HIP(hipMemcpy(homographies, dH, (size_t)F * 9 * sizeof(float), hipMemcpyDeviceToHost));
  return CC_OK;
}
