// cc_points.hip -- per-point kernels of the Calibrator surface: Distort / Undistort
// (/root/reference/src/calibrator.cpp:118-166). One thread per point.
// Compiled with -ffp-contract=off: the float/double promotion pattern of the reference's
// DistortNormalized (calibrator.cpp:70-83) is kept operation by operation.
#include "cc_common.hpp"

namespace cc {

struct PointParams {
  float fx, fy, px, py;      // K(0,0) K(1,1) K(0,2) K(1,2)
  float k1, k2, p1, p2, k3;  // dist_ = k1 k2 p1 p2 k3 (OpenCV order)
  float Ki[9];               // K^-1 (float, cofactor form as Eigen's 3x3 inverse)
};

// Calibrator::Distort -> DistortPixels<float,float> (calibrator.cpp:85-95,157-166)
__global__ void k_distort(PointParams P, int64_t n, const float2* __restrict__ xy, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xy[i].x, y = xy[i].y;
  const float r2 = x * x + y * y;
  const float r4 = r2 * r2;
  const float r6 = r4 * r2;
  const double r_mult = 1.0 + P.k1 * r2 + P.k2 * r4 + P.k3 * r6;  // calibrator.cpp:80
  const float nx = (float)(x * r_mult + 2.0 * P.p1 * x * y + P.p2 * (r2 + 2.0 * x * x));
  const float ny = (float)(y * r_mult + 2.0 * P.p2 * x * y + P.p1 * (r2 + 2.0 * y * y));
  out[i] = make_float2(P.fx * nx + P.px, P.fy * ny + P.py);
}

// Calibrator::Undistort (calibrator.cpp:118-155): the reference runs one 2-variable Ceres problem
// per point (DistortionError, DENSE_QR, <= 1000 iterations, start = distorted coordinates). Here:
// Levenberg-Marquardt with the same radius rules, iterated to the root of the distortion map.
__device__ __forceinline__ void distort_eval(double k1, double k2, double p1, double p2, double k3,
                                             double x, double y, double xd, double yd, double* r,
                                             double* J) {
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double m = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  r[0] = x * m + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x) - xd;
  r[1] = y * m + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y) - yd;
  const double mp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
  J[0] = m + 2.0 * mp * x * x + 2.0 * p1 * y + 6.0 * p2 * x;
  J[1] = J[2] = 2.0 * mp * x * y + 2.0 * p1 * x + 2.0 * p2 * y;
  J[3] = m + 2.0 * mp * y * y + 2.0 * p2 * x + 6.0 * p1 * y;
}

__global__ void k_undistort(PointParams P, int64_t n, const float2* __restrict__ uv, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float pxf = uv[i].x, pyf = uv[i].y;
  const float w = P.Ki[6] * pxf + P.Ki[7] * pyf + P.Ki[8];            // calibrator.cpp:136
  const double xd = (P.Ki[0] * pxf + P.Ki[1] * pyf + P.Ki[2]) / w;    // float expression -> double
  const double yd = (P.Ki[3] * pxf + P.Ki[4] * pyf + P.Ki[5]) / w;
  const double k1 = P.k1, k2 = P.k2, p1 = P.p1, p2 = P.p2, k3 = P.k3;
  double x = xd, y = yd, radius = 1e4, dec = 2.0;
  double r[2], J[4];
  distort_eval(k1, k2, p1, p2, k3, x, y, xd, yd, r, J);
  double cost = 0.5 * (r[0] * r[0] + r[1] * r[1]);
  for (int it = 0; it < 1000 && cost > 0.0; ++it) {
    const double h00 = J[0] * J[0] + J[2] * J[2], h01 = J[0] * J[1] + J[2] * J[3];
    const double h11 = J[1] * J[1] + J[3] * J[3];
    const double g0 = J[0] * r[0] + J[2] * r[1], g1 = J[1] * r[0] + J[3] * r[1];
    const double a00 = h00 + fmax(h00, 1e-6) / radius, a11 = h11 + fmax(h11, 1e-6) / radius;
    const double det = a00 * a11 - h01 * h01;
    const double dx = -(a11 * g0 - h01 * g1) / det, dy = -(a00 * g1 - h01 * g0) / det;
    if (!(fabs(dx) + fabs(dy) > 1e-17 * (fabs(x) + fabs(y) + 1e-300))) break;
    double rn[2], Jn[4];
    distort_eval(k1, k2, p1, p2, k3, x + dx, y + dy, xd, yd, rn, Jn);
    const double cn = 0.5 * (rn[0] * rn[0] + rn[1] * rn[1]);
    if (cn < cost) {
      x += dx; y += dy; cost = cn;
      r[0] = rn[0]; r[1] = rn[1];
      J[0] = Jn[0]; J[1] = Jn[1]; J[2] = Jn[2]; J[3] = Jn[3];
      radius = fmin(1e16, radius * 3.0);
      dec = 2.0;
    } else {
      radius /= dec; dec *= 2.0;
      if (radius < 1e-32) break;
    }
  }
  out[i] = make_float2((float)x, (float)y);
}

static void inv3f(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.0f / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

static int run_points(int device, const float* K, const float* dist, int64_t n, const float* in, float* out, bool undist) {
  if (n < 0 || n >= ((int64_t)1 << 39) || !K || !dist || (n > 0 && (!in || !out)))   // 32-bit launch grid of 256-thread blocks
    return fail(CC_ERR_BAD_ARGUMENT, "cc_(un)distort: bad arguments");
  if (int rc = select_device(device)) return rc;
  if (n == 0) return CC_OK;
  PointParams P;
  P.fx = K[0]; P.fy = K[4]; P.px = K[2]; P.py = K[5];
  P.k1 = dist[0]; P.k2 = dist[1]; P.p1 = dist[2]; P.p2 = dist[3]; P.k3 = dist[4];
  inv3f(K, P.Ki);
  // cached stream and cached scratch buffer (input | output): the Python workflow calls this once per frame
  // (cam_calibration.py:85-97), so the fixed cost matters more than the kernel
  hipStream_t stream = nullptr;
  if (int rc = stream_get(device, &stream)) return rc;
  float2* buf = nullptr;
  bool cached = false;
  const size_t bytes = (size_t)2 * n * sizeof(float2);
  struct Release {
    float2** p; int device; hipStream_t s; size_t bytes; bool* cached;
    ~Release() {
      const bool ok = hipStreamSynchronize(s) == hipSuccess;
      scratch_put(device, *p, bytes, *cached);
      if (ok) stream_put(device, s); else hipStreamDestroy(s);
    }
  } release{&buf, device, stream, bytes, &cached};
  {
    void* raw = nullptr;
    if (int rc = scratch_get(device, bytes, &raw, &cached)) return rc;
    buf = static_cast<float2*>(raw);
  }
  float2 *din = buf, *dout = buf + n;
  CC_HIP(hipMemcpyAsync(din, in, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, stream));
  const int threads = 256;
  const unsigned blocks = (unsigned)((n + threads - 1) / threads);
  if (undist) hipLaunchKernelGGL(k_undistort, dim3(blocks), dim3(threads), 0, stream, P, n, din, dout);
  else hipLaunchKernelGGL(k_distort, dim3(blocks), dim3(threads), 0, stream, P, n, din, dout);
  CC_HIP(hipGetLastError());
  CC_HIP(hipMemcpyAsync(out, dout, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, stream));
  CC_HIP(hipStreamSynchronize(stream));
  return CC_OK;
}

}  // namespace cc

extern "C" {
int cc_distort(int32_t device, const float* K9, const float* dist5, int64_t n, const float* xy, float* uv_out) {
  return cc::run_points(device, K9, dist5, n, xy, uv_out, false);
}
int cc_undistort(int32_t device, const float* K9, const float* dist5, int64_t n, const float* uv, float* xy_out) {
  return cc::run_points(device, K9, dist5, n, uv, xy_out, true);
}
}
