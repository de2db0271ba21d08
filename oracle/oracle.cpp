// oracle.cpp -- CPU restatement of the reference's reprojection-error LM path.
// TEST INFRASTRUCTURE ONLY (see oracle.h header comment for the parity status).
//
// Every function cites the reference lines it restates (paths relative to /root/reference).
// Build: oracle/Makefile (g++ -O2 -ffp-contract=off so float restatements keep their rounding).
#include "oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// small dense helpers
// ------------------------------------------------------------------------------------------

// In-place lower Cholesky of a row-major n x n SPD matrix. false if a pivot is not > 0 / finite.
bool cholesky(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return true;
}

// Solve L L^T x = b in place (L from cholesky()), b has `nrhs` columns, row-major n x nrhs.
void chol_solve(const double* L, int n, double* b, int nrhs) {
  for (int c = 0; c < nrhs; ++c) {
    for (int i = 0; i < n; ++i) {
      double s = b[i * nrhs + c];
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k * nrhs + c];
      b[i * nrhs + c] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = b[i * nrhs + c];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k * nrhs + c];
      b[i * nrhs + c] = s / L[i * n + i];
    }
  }
}

// Rotation matrix (row-major) of q/|q|, q = (w,x,y,z): what ceres::QuaternionRotatePoint
// applies (it normalises q first), used at calibrator.cpp:201 and extrinsics_calibrator.cpp:62,69.
void quat_to_R(const double* q, double* R) {
  const double n = 1.0 / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// ceres::QuaternionManifold::Plus (set at calibrator.cpp:298, extrinsics_calibrator.cpp:186,196):
// x_plus = [cos|d|, sin|d|/|d| * d] (x) x, Hamilton product, w first.
void quat_plus(const double* x, const double* d, double* out) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { for (int i = 0; i < 4; ++i) out[i] = x[i]; return; }
  const double s = std::sin(nd) / nd;
  const double a0 = std::cos(nd), a1 = s * d[0], a2 = s * d[1], a3 = s * d[2];
  out[0] = a0 * x[0] - a1 * x[1] - a2 * x[2] - a3 * x[3];
  out[1] = a0 * x[1] + a1 * x[0] + a2 * x[3] - a3 * x[2];
  out[2] = a0 * x[2] - a1 * x[3] + a2 * x[0] + a3 * x[1];
  out[3] = a0 * x[3] + a1 * x[2] - a2 * x[1] + a3 * x[0];
}

// Contribution of a pose block [q(4) t(3)] with tangent gradient g(6) to Ceres' gradient_max_norm,
// || x - Plus(x, -g) ||_inf (TrustRegionMinimizer: EvaluateGradientAndJacobian / the gradient-tolerance test; Plus is
// QuaternionManifold's, calibrator.cpp:298, extrinsics_calibrator.cpp:119,127). Written out for q - Plus(q, -g_rot) with
// 1 - cos|g| as a series of its own (Ceres subtracts two nearly equal quaternions); blocks with |g_rot| >= 1/4 report the
// tangent max-norm -- the HIP kernels follow the same rule (cc_common.hpp, pose_grad_proj_max), every decision is Ceres'
// for gradient tolerances below 0.14.
double pose_grad_proj_max(const double* q, const double* g) {
  const double gt = std::max(std::max(std::fabs(g[3]), std::fabs(g[4])), std::fabs(g[5]));
  const double n2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
  if (!(n2 < 0.0625)) return std::max(gt, std::max(std::max(std::fabs(g[0]), std::fabs(g[1])), std::fabs(g[2])));
  const double c1 = n2 * (1.0 / 2 + n2 * (-1.0 / 24 + n2 * (1.0 / 720 + n2 * (-1.0 / 40320 + n2 * (1.0 / 3628800 +
                    n2 * (-1.0 / 479001600 + n2 * (1.0 / 87178291200.0 + n2 * (-1.0 / 20922789888000.0))))))));
  const double s = 1.0 + n2 * (-1.0 / 6 + n2 * (1.0 / 120 + n2 * (-1.0 / 5040 + n2 * (1.0 / 362880 + n2 * (-1.0 / 39916800 +
                   n2 * (1.0 / 6227020800.0 + n2 * (-1.0 / 1307674368000.0 + n2 * (1.0 / 355687428096000.0))))))));
  const double w = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
  const double dw = c1 * w - s * (g[0] * v0 + g[1] * v1 + g[2] * v2);
  const double d0 = c1 * v0 + s * (w * g[0] + (g[1] * v2 - g[2] * v1));
  const double d1 = c1 * v1 + s * (w * g[1] + (g[2] * v0 - g[0] * v2));
  const double d2 = c1 * v2 + s * (w * g[2] + (g[0] * v1 - g[1] * v0));
  return std::max(std::max(gt, std::fabs(dw)), std::max(std::max(std::fabs(d0), std::fabs(d1)), std::fabs(d2)));
}

// ------------------------------------------------------------------------------------------
// residual models with analytic Jacobians
// ------------------------------------------------------------------------------------------

// ReprojectionError::operator() (calibrator.cpp:183-219) with DistortPixels/DistortNormalized
// (calibrator.cpp:70-95).  k = fx fy px py k1 k2 p1 p2 k3 (calibrator.cpp:168-179).
// J rows: d res / d [k(9), rot tangent(3), t(3)]; rot tangent is QuaternionManifold's delta
// (left-multiplied, rotation angle = 2|delta|), so d x_cam / d delta = -2 [R X]_x.
inline void intr_eval(const double* k, const double* R, const double* t, const double* X,
                      double u, double v, double* res, double (*J)[15]) {
  const double a0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double a1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double a2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double xc = a0 + t[0], yc = a1 + t[1], zc = a2 + t[2];
  const double iz = 1.0 / zc;
  const double x = xc * iz, y = yc * iz;
  const double fx = k[0], fy = k[1], px = k[2], py = k[3];
  const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7], k3 = k[8];
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double m = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  const double xd = x * m + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
  const double yd = y * m + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y);
  res[0] = fx * xd + px - u;
  res[1] = fy * yd + py - v;
  if (!J) return;
  J[0][0] = xd; J[0][1] = 0;  J[0][2] = 1; J[0][3] = 0;
  J[0][4] = fx * x * r2; J[0][5] = fx * x * r4; J[0][6] = fx * 2.0 * x * y;
  J[0][7] = fx * (r2 + 2.0 * x * x); J[0][8] = fx * x * r6;
  J[1][0] = 0;  J[1][1] = yd; J[1][2] = 0; J[1][3] = 1;
  J[1][4] = fy * y * r2; J[1][5] = fy * y * r4; J[1][6] = fy * (r2 + 2.0 * y * y);
  J[1][7] = fy * 2.0 * x * y; J[1][8] = fy * y * r6;
  const double mp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;  // dm/d(r2)
  const double dxx = m + 2.0 * mp * x * x + 2.0 * p1 * y + 6.0 * p2 * x;
  const double dxy = 2.0 * mp * x * y + 2.0 * p1 * x + 2.0 * p2 * y;
  const double dyy = m + 2.0 * mp * y * y + 2.0 * p2 * x + 6.0 * p1 * y;
  double B[2][3];
  B[0][0] = fx * dxx * iz; B[0][1] = fx * dxy * iz; B[0][2] = -(fx * dxx * x + fx * dxy * y) * iz;
  B[1][0] = fy * dxy * iz; B[1][1] = fy * dyy * iz; B[1][2] = -(fy * dxy * x + fy * dyy * y) * iz;
  for (int i = 0; i < 2; ++i) {
    J[i][9]  = 2.0 * (B[i][2] * a1 - B[i][1] * a2);
    J[i][10] = 2.0 * (B[i][0] * a2 - B[i][2] * a0);
    J[i][11] = 2.0 * (B[i][1] * a0 - B[i][0] * a1);
    J[i][12] = B[i][0]; J[i][13] = B[i][1]; J[i][14] = B[i][2];
  }
}

// ReprojectionErrorExtrinsics::operator() (extrinsics_calibrator.cpp:51-84).
// J rows: d res / d [cam rot(3), cam t(3), frame rot(3), frame t(3)].
inline void rig_eval(const double* Rf, const double* tf, const double* Rc, const double* tc,
                     const double* X, double u, double v, double* res, double (*J)[12]) {
  const double b0 = Rf[0] * X[0] + Rf[1] * X[1] + Rf[2] * X[2];
  const double b1 = Rf[3] * X[0] + Rf[4] * X[1] + Rf[5] * X[2];
  const double b2 = Rf[6] * X[0] + Rf[7] * X[1] + Rf[8] * X[2];
  const double r0 = b0 + tf[0], r1 = b1 + tf[1], r2 = b2 + tf[2];  // X_rig
  const double a0 = Rc[0] * r0 + Rc[1] * r1 + Rc[2] * r2;
  const double a1 = Rc[3] * r0 + Rc[4] * r1 + Rc[5] * r2;
  const double a2 = Rc[6] * r0 + Rc[7] * r1 + Rc[8] * r2;
  const double xc = a0 + tc[0], yc = a1 + tc[1], zc = a2 + tc[2];
  const double iz = 1.0 / zc;
  const double x = xc * iz, y = yc * iz;
  res[0] = x - u;
  res[1] = y - v;
  if (!J) return;
  const double B[2][3] = {{iz, 0.0, -x * iz}, {0.0, iz, -y * iz}};
  for (int i = 0; i < 2; ++i) {
    J[i][0] = 2.0 * (B[i][2] * a1 - B[i][1] * a2);
    J[i][1] = 2.0 * (B[i][0] * a2 - B[i][2] * a0);
    J[i][2] = 2.0 * (B[i][1] * a0 - B[i][0] * a1);
    J[i][3] = B[i][0]; J[i][4] = B[i][1]; J[i][5] = B[i][2];
    // M = B * Rc (2x3), d res / d X_rig
    const double m0 = B[i][0] * Rc[0] + B[i][1] * Rc[3] + B[i][2] * Rc[6];
    const double m1 = B[i][0] * Rc[1] + B[i][1] * Rc[4] + B[i][2] * Rc[7];
    const double m2 = B[i][0] * Rc[2] + B[i][1] * Rc[5] + B[i][2] * Rc[8];
    J[i][6] = 2.0 * (m2 * b1 - m1 * b2);
    J[i][7] = 2.0 * (m0 * b2 - m2 * b0);
    J[i][8] = 2.0 * (m1 * b0 - m0 * b1);
    J[i][9] = m0; J[i][10] = m1; J[i][11] = m2;
  }
}

// EXTENSION (no counterpart in the reference): the rig model composed with the
// pixel model -- x_cam = R_c (R_f X + t_f) + t_c as in ReprojectionErrorExtrinsics
// (extrinsics_calibrator.cpp:51-84), then DistortNormalized/DistortPixels with 9 intrinsics shared by
// all cameras (calibrator.cpp:70-95); residual in pixels.
// J rows: d res / d [cam rot(3), cam t(3), frame rot(3), frame t(3), k(9)].
inline void rigk_eval(const double* k, const double* Rf, const double* tf, const double* Rc, const double* tc,
                      const double* X, double u, double v, double* res, double (*J)[21]) {
  const double b0 = Rf[0] * X[0] + Rf[1] * X[1] + Rf[2] * X[2];
  const double b1 = Rf[3] * X[0] + Rf[4] * X[1] + Rf[5] * X[2];
  const double b2 = Rf[6] * X[0] + Rf[7] * X[1] + Rf[8] * X[2];
  const double r0 = b0 + tf[0], r1 = b1 + tf[1], r2_ = b2 + tf[2];  // X_rig
  const double a0 = Rc[0] * r0 + Rc[1] * r1 + Rc[2] * r2_;
  const double a1 = Rc[3] * r0 + Rc[4] * r1 + Rc[5] * r2_;
  const double a2 = Rc[6] * r0 + Rc[7] * r1 + Rc[8] * r2_;
  const double xc = a0 + tc[0], yc = a1 + tc[1], zc = a2 + tc[2];
  const double iz = 1.0 / zc;
  const double x = xc * iz, y = yc * iz;
  const double fx = k[0], fy = k[1], px = k[2], py = k[3];
  const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7], k3 = k[8];
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double m = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  const double xd = x * m + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
  const double yd = y * m + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y);
  res[0] = fx * xd + px - u;
  res[1] = fy * yd + py - v;
  if (!J) return;
  J[0][12] = xd; J[0][13] = 0;  J[0][14] = 1; J[0][15] = 0;
  J[0][16] = fx * x * r2; J[0][17] = fx * x * r4; J[0][18] = fx * 2.0 * x * y;
  J[0][19] = fx * (r2 + 2.0 * x * x); J[0][20] = fx * x * r6;
  J[1][12] = 0;  J[1][13] = yd; J[1][14] = 0; J[1][15] = 1;
  J[1][16] = fy * y * r2; J[1][17] = fy * y * r4; J[1][18] = fy * (r2 + 2.0 * y * y);
  J[1][19] = fy * 2.0 * x * y; J[1][20] = fy * y * r6;
  const double mp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
  const double dxx = m + 2.0 * mp * x * x + 2.0 * p1 * y + 6.0 * p2 * x;
  const double dxy = 2.0 * mp * x * y + 2.0 * p1 * x + 2.0 * p2 * y;
  const double dyy = m + 2.0 * mp * y * y + 2.0 * p2 * x + 6.0 * p1 * y;
  double B[2][3];  // d res / d x_cam
  B[0][0] = fx * dxx * iz; B[0][1] = fx * dxy * iz; B[0][2] = -(fx * dxx * x + fx * dxy * y) * iz;
  B[1][0] = fy * dxy * iz; B[1][1] = fy * dyy * iz; B[1][2] = -(fy * dxy * x + fy * dyy * y) * iz;
  for (int i = 0; i < 2; ++i) {
    J[i][0] = 2.0 * (B[i][2] * a1 - B[i][1] * a2);
    J[i][1] = 2.0 * (B[i][0] * a2 - B[i][2] * a0);
    J[i][2] = 2.0 * (B[i][1] * a0 - B[i][0] * a1);
    J[i][3] = B[i][0]; J[i][4] = B[i][1]; J[i][5] = B[i][2];
    const double m0 = B[i][0] * Rc[0] + B[i][1] * Rc[3] + B[i][2] * Rc[6];
    const double m1 = B[i][0] * Rc[1] + B[i][1] * Rc[4] + B[i][2] * Rc[7];
    const double m2 = B[i][0] * Rc[2] + B[i][1] * Rc[5] + B[i][2] * Rc[8];
    J[i][6] = 2.0 * (m2 * b1 - m1 * b2);
    J[i][7] = 2.0 * (m0 * b2 - m2 * b0);
    J[i][8] = 2.0 * (m1 * b0 - m0 * b1);
    J[i][9] = m0; J[i][10] = m1; J[i][11] = m2;
  }
}

// ------------------------------------------------------------------------------------------
// Block-arrow LM (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy semantics,
// exact Schur linear solve).  Shared block of tangent size S, F per-frame blocks of size 6.
// ------------------------------------------------------------------------------------------

struct Blocks {
  std::vector<double> Hpp, Hps, gp, Hss, gs;  // F*36, F*6*S, F*6, S*S, S
  void resize(int64_t F, int S) {
    Hpp.assign(F * 36, 0); Hps.assign(F * 6 * S, 0); gp.assign(F * 6, 0);
    Hss.assign((size_t)S * S, 0); gs.assign(S, 0);
  }
};

struct ArrowProblem {
  int64_t F = 0;
  int S = 0;      // shared tangent size
  int S_amb = 0;  // shared ambient size
  std::vector<uint8_t> shared_fixed;       // S: tangent coordinate not optimised
  std::vector<uint8_t> shared_amb_active;  // S_amb: parameter counts in |x| and |step|
  std::vector<uint8_t> frame_active;       // F
  int64_t shared_pose_blocks = 0;          // leading [q t] blocks of the shared parameters (rig cameras): 7 ambient, 6 tangent each
  int num_threads = 1;
  virtual ~ArrowProblem() {}
  // Returns local cost; fills blocks if B != nullptr (all of B is overwritten).
  virtual double eval(const double* shared, const double* fq, const double* ft, Blocks* B) = 0;
  virtual void plus_shared(const double* shared, const double* delta, double* out) = 0;
};

// Ceres TrustRegionStepEvaluator (non-monotonic acceptance, Conn/Gould/Toint alg. 10.1.2).
struct StepEvaluator {
  int max_nonmono;
  double minimum_cost, current_cost, reference_cost, candidate_cost;
  double acc_ref = 0, acc_cand = 0;
  int num_nonmono = 0;
  StepEvaluator(double c0, int maxn)
      : max_nonmono(maxn), minimum_cost(c0), current_cost(c0), reference_cost(c0), candidate_cost(c0) {}
  double quality(double cost, double mcc) const {
    if (!(cost < std::numeric_limits<double>::max())) return std::numeric_limits<double>::lowest();
    const double rel = (current_cost - cost) / mcc;
    const double hist = (reference_cost - cost) / (acc_ref + mcc);
    return std::max(rel, hist);
  }
  void accepted(double cost, double mcc) {
    current_cost = cost;
    acc_cand += mcc;
    acc_ref += mcc;
    if (current_cost < minimum_cost) {
      minimum_cost = current_cost;
      num_nonmono = 0;
      candidate_cost = current_cost;
      acc_cand = 0;
    } else {
      ++num_nonmono;
      if (current_cost > candidate_cost) { candidate_cost = current_cost; acc_cand = 0; }
    }
    if (num_nonmono == max_nonmono) { reference_cost = candidate_cost; acc_ref = acc_cand; }
  }
};

void no_allreduce(void*, double*, int32_t, int32_t) {}

int run_lm(ArrowProblem& P, const oc_options& o, double* shared, double* fq, double* ft,
           oc_summary* summary, oc_allreduce_fn ar, void* ctx) {
  const auto t_start = std::chrono::steady_clock::now();
  if (!ar) ar = no_allreduce;
  const int64_t F = P.F;
  const int S = P.S, SA = P.S_amb;
  Blocks B, Bc;
  B.resize(F, S);
  Bc.resize(F, S);
  std::vector<double> red;  // scratch for all-reduces

  auto eval_global = [&](const double* sh, const double* q, const double* t, Blocks& out) {
    double cost = P.eval(sh, q, t, &out);
    red.assign((size_t)S * S + S + 1, 0);
    std::copy(out.Hss.begin(), out.Hss.end(), red.begin());
    std::copy(out.gs.begin(), out.gs.end(), red.begin() + (size_t)S * S);
    red[(size_t)S * S + S] = cost;
    ar(ctx, red.data(), (int32_t)red.size(), 0);
    std::copy(red.begin(), red.begin() + (size_t)S * S, out.Hss.begin());
    std::copy(red.begin() + (size_t)S * S, red.begin() + (size_t)S * S + S, out.gs.begin());
    return red[(size_t)S * S + S];
  };
  auto grad_max = [&](const Blocks& b) {
    double g = 0;
    for (int64_t f = 0; f < F; ++f)
      if (P.frame_active[f]) g = std::max(g, pose_grad_proj_max(&fq[f * 4], &b.gp[f * 6]));
    ar(ctx, &g, 1, 1);
    for (int64_t c = 0; c < P.shared_pose_blocks; ++c)
      if (!P.shared_fixed[c * 6]) g = std::max(g, pose_grad_proj_max(&shared[c * 7], &b.gs[c * 6]));
    for (int i = (int)(6 * P.shared_pose_blocks); i < S; ++i)
      if (!P.shared_fixed[i]) g = std::max(g, std::fabs(b.gs[i]));
    return g;
  };
  auto x_norm = [&](const double* sh, const double* q, const double* t) {
    double n = 0;
    for (int64_t f = 0; f < F; ++f)
      if (P.frame_active[f]) {
        for (int i = 0; i < 4; ++i) n += q[f * 4 + i] * q[f * 4 + i];
        for (int i = 0; i < 3; ++i) n += t[f * 3 + i] * t[f * 3 + i];
      }
    ar(ctx, &n, 1, 0);
    for (int i = 0; i < SA; ++i)
      if (P.shared_amb_active[i]) n += sh[i] * sh[i];
    return std::sqrt(n);
  };

  double x_cost = eval_global(shared, fq, ft, B);

  // Jacobi scaling, computed once from the initial Jacobian (Ceres trust_region_minimizer.cc):
  // scale_i = 1 / (1 + sqrt(sum_rows J_ri^2)).
  std::vector<double> ss(S, 1.0), sp(F * 6, 1.0);
  if (o.jacobi_scaling) {
    for (int i = 0; i < S; ++i) ss[i] = 1.0 / (1.0 + std::sqrt(B.Hss[(size_t)i * S + i]));
    for (int64_t f = 0; f < F; ++f)
      for (int i = 0; i < 6; ++i) sp[f * 6 + i] = 1.0 / (1.0 + std::sqrt(B.Hpp[f * 36 + i * 6 + i]));
  }

  StepEvaluator ev(x_cost, o.use_nonmonotonic_steps ? o.max_consecutive_nonmonotonic_steps : 0);
  double radius = o.initial_radius, decrease_factor = 2.0;
  double xn = x_norm(shared, fq, ft);
  double gmax = grad_max(B);
  int n_invalid = 0, iters = 0, n_success = 0, term = OC_NO_CONVERGENCE;
  const double initial_cost = x_cost;
  int log_len = 0;
  auto log = [&](double cost, double cc, double mcc, double rd, double sn, int acc, int valid) {
    if (summary && summary->log && log_len < summary->log_capacity) {
      oc_iteration& it = summary->log[log_len];
      it.cost = cost; it.cost_change = cc; it.model_cost_change = mcc; it.relative_decrease = rd;
      it.gradient_max_norm = gmax; it.step_norm = sn; it.radius = radius; it.accepted = acc;
      it.valid = valid;
    }
    ++log_len;
  };

  std::vector<double> Y(F * 6 * (S + 1)), Ssum((size_t)S * S + S), Sred((size_t)S * S), bred(S);
  std::vector<double> ds(S), dp(F * 6), cand_sh(SA), cand_q(F * 4), cand_t(F * 3);

  if (gmax <= o.gradient_tolerance) term = OC_CONVERGENCE_GRADIENT;

  while (term == OC_NO_CONVERGENCE && iters < o.max_iterations) {
    if (radius < o.min_radius) { term = OC_MIN_RADIUS; break; }
    ++iters;
    // ---- LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system, via Schur ----
    bool ok = true;
    std::fill(Ssum.begin(), Ssum.end(), 0.0);
    const int W = S + 1;
    for (int64_t f = 0; f < F && ok; ++f) {
      if (!P.frame_active[f]) { std::fill(&Y[f * 6 * W], &Y[(f + 1) * 6 * W], 0.0); continue; }
      double A[36];
      const double* s6 = &sp[f * 6];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) A[i * 6 + j] = s6[i] * B.Hpp[f * 36 + i * 6 + j] * s6[j];
      for (int i = 0; i < 6; ++i)
        A[i * 6 + i] += std::min(std::max(A[i * 6 + i], o.min_lm_diagonal), o.max_lm_diagonal) / radius;
      if (!cholesky(A, 6)) { ok = false; break; }
      double* Yf = &Y[f * 6 * W];
      std::vector<double> Wm(6 * W);
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < S; ++j) Wm[i * W + j] = s6[i] * B.Hps[(f * 6 + i) * S + j] * ss[j];
        Wm[i * W + S] = s6[i] * B.gp[f * 6 + i];
      }
      std::copy(Wm.begin(), Wm.end(), Yf);
      chol_solve(A, 6, Yf, W);
      for (int j = 0; j < S; ++j) {
        for (int k = 0; k < S; ++k) {
          double a = 0;
          for (int i = 0; i < 6; ++i) a += Wm[i * W + j] * Yf[i * W + k];
          Ssum[(size_t)j * S + k] += a;
        }
        double a = 0;
        for (int i = 0; i < 6; ++i) a += Wm[i * W + j] * Yf[i * W + S];
        Ssum[(size_t)S * S + j] += a;
      }
    }
    {
      double flag = ok ? 0.0 : 1.0;
      ar(ctx, &flag, 1, 1);
      ok = flag == 0.0;
    }
    ar(ctx, Ssum.data(), (int32_t)Ssum.size(), 0);
    if (ok) {
      for (int i = 0; i < S; ++i) {
        for (int j = 0; j < S; ++j)
          Sred[(size_t)i * S + j] = ss[i] * B.Hss[(size_t)i * S + j] * ss[j] - Ssum[(size_t)i * S + j];
        const double hii = ss[i] * B.Hss[(size_t)i * S + i] * ss[i];
        Sred[(size_t)i * S + i] += std::min(std::max(hii, o.min_lm_diagonal), o.max_lm_diagonal) / radius;
        bred[i] = ss[i] * B.gs[i] - Ssum[(size_t)S * S + i];
      }
      for (int i = 0; i < S; ++i)
        if (P.shared_fixed[i]) {
          for (int j = 0; j < S; ++j) Sred[(size_t)i * S + j] = Sred[(size_t)j * S + i] = 0;
          Sred[(size_t)i * S + i] = 1;
          bred[i] = 0;
        }
      if (S > 0) {
        ok = cholesky(Sred.data(), S);
        if (ok) {
          chol_solve(Sred.data(), S, bred.data(), 1);
          for (int i = 0; i < S; ++i) ds[i] = -bred[i];  // scaled shared step
        }
      }
    }
    double mcc = 0, cand_cost = 0, step_norm = 0, quality = 0;
    if (ok) {
      double mloc = 0;
      for (int64_t f = 0; f < F; ++f) {
        const double* Yf = &Y[f * 6 * W];
        double d6[6];
        for (int i = 0; i < 6; ++i) {
          double a = Yf[i * W + S];
          for (int j = 0; j < S; ++j) a += Yf[i * W + j] * ds[j];
          d6[i] = -a * sp[f * 6 + i];  // unscaled pose step
          dp[f * 6 + i] = d6[i];
        }
        if (!P.frame_active[f]) continue;
        for (int i = 0; i < 6; ++i) {
          double hd = 0;
          for (int j = 0; j < 6; ++j) hd += B.Hpp[f * 36 + i * 6 + j] * d6[j];
          double cs = 0;
          for (int j = 0; j < S; ++j) cs += B.Hps[(f * 6 + i) * S + j] * (ds[j] * ss[j]);
          mloc += d6[i] * (B.gp[f * 6 + i] + 0.5 * hd + cs);
        }
      }
      ar(ctx, &mloc, 1, 0);
      for (int i = 0; i < S; ++i) ds[i] *= ss[i];  // unscale
      double msh = 0;
      for (int i = 0; i < S; ++i) {
        double hd = 0;
        for (int j = 0; j < S; ++j) hd += B.Hss[(size_t)i * S + j] * ds[j];
        msh += ds[i] * (B.gs[i] + 0.5 * hd);
      }
      mcc = -(mloc + msh);
      ok = std::isfinite(mcc) && mcc > 0.0;
    }
    if (!ok) {
      // TrustRegionMinimizer::HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid
      ++n_invalid;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      log(x_cost, 0, mcc, 0, 0, 0, 0);
      if (n_invalid >= o.max_consecutive_invalid_steps) { term = OC_FAILURE_INVALID_STEPS; break; }
      continue;
    }
    n_invalid = 0;
    // ---- candidate point ----
    P.plus_shared(shared, ds.data(), cand_sh.data());
    double sn2 = 0;
    for (int64_t f = 0; f < F; ++f) {
      quat_plus(&fq[f * 4], &dp[f * 6], &cand_q[f * 4]);
      for (int i = 0; i < 3; ++i) cand_t[f * 3 + i] = ft[f * 3 + i] + dp[f * 6 + 3 + i];
      if (!P.frame_active[f]) continue;
      for (int i = 0; i < 4; ++i) { const double d = cand_q[f * 4 + i] - fq[f * 4 + i]; sn2 += d * d; }
      for (int i = 0; i < 3; ++i) { const double d = cand_t[f * 3 + i] - ft[f * 3 + i]; sn2 += d * d; }
    }
    ar(ctx, &sn2, 1, 0);
    for (int i = 0; i < SA; ++i)
      if (P.shared_amb_active[i]) { const double d = cand_sh[i] - shared[i]; sn2 += d * d; }
    step_norm = std::sqrt(sn2);
    cand_cost = eval_global(cand_sh.data(), cand_q.data(), cand_t.data(), Bc);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    // ParameterToleranceReached / FunctionToleranceReached (checked before acceptance, x stays)
    if (step_norm <= o.parameter_tolerance * (xn + o.parameter_tolerance)) {
      term = OC_CONVERGENCE_PARAMETER;
      log(x_cost, x_cost - cand_cost, mcc, 0, step_norm, 0, 1);
      break;
    }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= o.function_tolerance * x_cost) {
      term = OC_CONVERGENCE_FUNCTION;
      log(x_cost, cost_change, mcc, 0, step_norm, 0, 1);
      break;
    }
    quality = ev.quality(cand_cost, mcc);
    if (quality > o.min_relative_decrease) {
      // HandleSuccessfulStep
      std::copy(cand_sh.begin(), cand_sh.end(), shared);
      std::copy(cand_q.begin(), cand_q.end(), fq);
      std::copy(cand_t.begin(), cand_t.end(), ft);
      std::swap(B, Bc);
      x_cost = cand_cost;
      xn = x_norm(shared, fq, ft);
      gmax = grad_max(B);
      const double q3 = 2.0 * quality - 1.0;
      radius = radius / std::max(1.0 / 3.0, 1.0 - q3 * q3 * q3);
      radius = std::min(o.max_radius, radius);
      decrease_factor = 2.0;
      ev.accepted(cand_cost, mcc);
      ++n_success;
      log(x_cost, cost_change, mcc, quality, step_norm, 1, 1);
      // TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue: max iterations is tested before
      // the gradient tolerance, the minimum radius after it (top of the loop)
      if (iters < o.max_iterations && gmax <= o.gradient_tolerance) { term = OC_CONVERGENCE_GRADIENT; break; }
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      log(x_cost, cost_change, mcc, quality, step_norm, 0, 1);
    }
  }
  if (summary) {
    summary->iterations = iters;
    summary->successful_steps = n_success;
    summary->termination = term;
    summary->log_len = std::min(log_len, summary->log ? summary->log_capacity : 0);
    summary->initial_cost = initial_cost;
    summary->final_cost = x_cost;
    summary->seconds =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  }
  return 0;
}

// run fn(f0, f1, tid) over [0,F) split in nthreads contiguous ranges
template <class Fn>
void parallel_frames(int64_t F, int nthreads, Fn fn) {
  nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, F));
  if (nthreads == 1) { fn(0, F, 0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) {
    const int64_t f0 = F * t / nthreads, f1 = F * (t + 1) / nthreads;
    th.emplace_back([=] { fn(f0, f1, t); });
  }
  for (auto& x : th) x.join();
}

// ------------------------------------------------------------------------------------------
// Intrinsics problem (Calibrator::Optimize, calibrator.cpp:221-336)
// ------------------------------------------------------------------------------------------

// Accumulate the 16x16 Gram block of one frame (upper triangle, row-major 16x16 with both
// triangles filled on return). v = [J_intr(9) J_pose(6) r].
double intr_frame_block(const double* intr, uint32_t mask, const double* q, const double* t,
                        const float* uv, const float* xyz, int64_t n, double* G /*256*/) {
  double R[9];
  quat_to_R(q, R);
  double acc[256];
  std::fill(acc, acc + 256, 0.0);
  for (int64_t i = 0; i < n; ++i) {
    // float -> double exactly as calibrator.cpp:292,294
    const double X[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    double res[2], J[2][15];
    intr_eval(intr, R, t, X, uv[i * 2], uv[i * 2 + 1], res, J);
    for (int r = 0; r < 2; ++r) {
      double v[16];
      for (int c = 0; c < 15; ++c) v[c] = J[r][c];
      for (int c = 0; c < 9; ++c)
        if (mask & (1u << c)) v[c] = 0.0;  // SubsetManifold (calibrator.cpp:305-312)
      v[15] = res[r];
      for (int a = 0; a < 16; ++a)
        for (int b = a; b < 16; ++b) acc[a * 16 + b] += v[a] * v[b];
    }
  }
  for (int a = 0; a < 16; ++a)
    for (int b = a; b < 16; ++b) G[a * 16 + b] = G[b * 16 + a] = acc[a * 16 + b];
  return 0.5 * acc[255];
}

struct IntrinsicsProblem : ArrowProblem {
  const int64_t* off;
  const float* uv;
  const float* xyz;
  uint32_t mask;
  double eval(const double* shared, const double* fq, const double* ft, Blocks* B) override {
    const int nt = std::max(1, num_threads);
    std::vector<double> cost_t(nt, 0.0);
    std::vector<std::vector<double>> hs(nt, std::vector<double>(9 * 9 + 9, 0.0));
    if (B) B->resize(F, 9);
    parallel_frames(F, nt, [&](int64_t f0, int64_t f1, int tid) {
      double G[256];
      double c = 0;
      for (int64_t f = f0; f < f1; ++f) {
        const int64_t s = off[f], n = off[f + 1] - off[f];
        c += intr_frame_block(shared, mask, &fq[f * 4], &ft[f * 3], uv + s * 2, xyz + s * 3, n, G);
        if (!B) continue;
        for (int i = 0; i < 6; ++i) {
          for (int j = 0; j < 6; ++j) B->Hpp[f * 36 + i * 6 + j] = G[(9 + i) * 16 + 9 + j];
          for (int j = 0; j < 9; ++j) B->Hps[(f * 6 + i) * 9 + j] = G[(9 + i) * 16 + j];
          B->gp[f * 6 + i] = G[(9 + i) * 16 + 15];
        }
        for (int i = 0; i < 9; ++i) {
          for (int j = 0; j < 9; ++j) hs[tid][i * 9 + j] += G[i * 16 + j];
          hs[tid][81 + i] += G[i * 16 + 15];
        }
      }
      cost_t[tid] = c;
    });
    double cost = 0;
    for (int t = 0; t < nt; ++t) {
      cost += cost_t[t];
      if (B) {
        for (int i = 0; i < 81; ++i) B->Hss[i] += hs[t][i];
        for (int i = 0; i < 9; ++i) B->gs[i] += hs[t][81 + i];
      }
    }
    return cost;
  }
  void plus_shared(const double* shared, const double* delta, double* out) override {
    for (int i = 0; i < 9; ++i) out[i] = shared[i] + ((mask & (1u << i)) ? 0.0 : delta[i]);
  }
};

int intrinsics_solve_impl(const oc_options* opt, int64_t F, const int64_t* off, const float* uv,
                          const float* xyz, double* intr, uint32_t mask, double* q, double* t,
                          oc_summary* summary, oc_allreduce_fn ar, void* ctx) {
  oc_options o;
  if (opt) o = *opt; else oc_options_init(&o);
  IntrinsicsProblem P;
  P.F = F; P.S = 9; P.S_amb = 9;
  P.off = off; P.uv = uv; P.xyz = xyz; P.mask = mask;
  P.num_threads = o.num_threads;
  P.shared_fixed.resize(9);
  for (int i = 0; i < 9; ++i) P.shared_fixed[i] = (mask >> i) & 1;
  P.shared_amb_active.assign(9, 1);
  P.frame_active.assign(F, 1);
  return run_lm(P, o, intr, q, t, summary, ar, ctx);
}

// ------------------------------------------------------------------------------------------
// Rig problem (ExtrinsicsCalibrator::Optimize, extrinsics_calibrator.cpp:86-257)
// ------------------------------------------------------------------------------------------

// ceres::HuberLoss(a) + Corrector: rho(s) = s (s <= a^2) else 2 a sqrt(s) - a^2; rho'' <= 0
// always, so residual and Jacobian are both scaled by sqrt(rho').
inline void huber(double a, double s, double* rho, double* sqrt_rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    *rho = 2.0 * a * r - b;
    *sqrt_rho1 = std::sqrt(std::max(std::numeric_limits<double>::min(), a / r));
  } else {
    *rho = s;
    *sqrt_rho1 = 1.0;
  }
}

struct RigProblem : ArrowProblem {
  int64_t C;
  const int64_t* off;
  const uint32_t* ocam;
  const uint64_t* oworld;
  const float* ouv;
  const float* wxyz;
  double huber_a;
  std::vector<uint8_t> cam_fixed;  // frozen or unobserved
  double* obs_cost = nullptr;      // optional per-observation 1/2 rho output

  double eval(const double* shared, const double* fq, const double* ft, Blocks* B) override {
    const int Sd = S;
    const int nt = std::max(1, num_threads);
    std::vector<double> Rc(C * 9);
    for (int64_t c = 0; c < C; ++c) quat_to_R(&shared[c * 7], &Rc[c * 9]);
    std::vector<double> cost_t(nt, 0.0);
    std::vector<std::vector<double>> hs(nt);
    if (B) { B->resize(F, Sd); for (auto& h : hs) h.assign((size_t)Sd * Sd + Sd, 0.0); }
    parallel_frames(F, nt, [&](int64_t f0, int64_t f1, int tid) {
      double c = 0;
      for (int64_t f = f0; f < f1; ++f) {
        double Rf[9];
        quat_to_R(&fq[f * 4], Rf);
        for (int64_t k = off[f]; k < off[f + 1]; ++k) {
          const uint32_t cam = ocam[k];
          const float* Xf = &wxyz[oworld[k] * 3];
          const double X[3] = {Xf[0], Xf[1], Xf[2]};  // extrinsics_calibrator.cpp:136
          double res[2], J[2][12];
          rig_eval(Rf, &ft[f * 3], &Rc[cam * 9], &shared[cam * 7 + 4], X, ouv[k * 2], ouv[k * 2 + 1],
                   res, B ? J : nullptr);
          double rho, sr;
          huber(huber_a, res[0] * res[0] + res[1] * res[1], &rho, &sr);
          c += 0.5 * rho;
          if (obs_cost) obs_cost[k] = 0.5 * rho;
          if (!B) continue;
          const bool fixed = cam_fixed[cam];
          double* hss = hs[tid].data();
          for (int r = 0; r < 2; ++r) {
            double vc[6], vf[6];
            for (int i = 0; i < 6; ++i) { vc[i] = fixed ? 0.0 : sr * J[r][i]; vf[i] = sr * J[r][6 + i]; }
            const double rr = sr * res[r];
            for (int i = 0; i < 6; ++i) {
              for (int j = 0; j < 6; ++j) {
                B->Hpp[f * 36 + i * 6 + j] += vf[i] * vf[j];
                B->Hps[(f * 6 + i) * Sd + cam * 6 + j] += vf[i] * vc[j];
                hss[(size_t)(cam * 6 + i) * Sd + cam * 6 + j] += vc[i] * vc[j];
              }
              B->gp[f * 6 + i] += vf[i] * rr;
              hss[(size_t)Sd * Sd + cam * 6 + i] += vc[i] * rr;
            }
          }
        }
      }
      cost_t[tid] = c;
    });
    double cost = 0;
    for (int t = 0; t < nt; ++t) {
      cost += cost_t[t];
      if (B) {
        for (size_t i = 0; i < (size_t)Sd * Sd; ++i) B->Hss[i] += hs[t][i];
        for (int i = 0; i < Sd; ++i) B->gs[i] += hs[t][(size_t)Sd * Sd + i];
      }
    }
    return cost;
  }
  void plus_shared(const double* shared, const double* delta, double* out) override {
    for (int64_t c = 0; c < C; ++c) {
      if (cam_fixed[c]) {
        for (int i = 0; i < 7; ++i) out[c * 7 + i] = shared[c * 7 + i];
        continue;
      }
      quat_plus(&shared[c * 7], &delta[c * 6], &out[c * 7]);
      for (int i = 0; i < 3; ++i) out[c * 7 + 4 + i] = shared[c * 7 + 4 + i] + delta[c * 6 + 3 + i];
    }
  }
};

// EXTENSION: rig poses + intrinsics, either 9 shared by all cameras (NK = 1) or 9 per camera (NK = C).
// Shared tangent = [cam 0 (6) ... cam C-1 (6) | k set 0 (9) ... k set NK-1 (9)], ambient = [cam (7 each) | k sets];
// pixel observations; Huber on the pixel residual (a <= 0: off).
struct RigKProblem : ArrowProblem {
  int64_t NK = 1;                   // intrinsics sets
  std::vector<uint32_t> kmasks;     // NK: bit i = intrinsic i of the set is held constant
  int64_t kset(uint32_t cam) const { return NK == 1 ? 0 : (int64_t)cam; }
  int64_t C;
  const int64_t* off;
  const uint32_t* ocam;
  const uint64_t* oworld;
  const float* ouv;
  const float* wxyz;
  double huber_a;
  std::vector<uint8_t> cam_fixed;
  double* obs_cost = nullptr;

  double eval(const double* shared, const double* fq, const double* ft, Blocks* B) override {
    const int Sd = S;
    const int K0 = (int)(6 * C);
    const int nt = std::max(1, num_threads);
    std::vector<double> Rc(C * 9);
    for (int64_t c = 0; c < C; ++c) quat_to_R(&shared[c * 7], &Rc[c * 9]);
    std::vector<double> cost_t(nt, 0.0);
    std::vector<std::vector<double>> hs(nt);
    if (B) { B->resize(F, Sd); for (auto& h : hs) h.assign((size_t)Sd * Sd + Sd, 0.0); }
    parallel_frames(F, nt, [&](int64_t f0, int64_t f1, int tid) {
      double c = 0;
      for (int64_t f = f0; f < f1; ++f) {
        double Rf[9];
        quat_to_R(&fq[f * 4], Rf);
        for (int64_t o = off[f]; o < off[f + 1]; ++o) {
          const uint32_t cam = ocam[o];
          const float* Xf = &wxyz[oworld[o] * 3];
          const double X[3] = {Xf[0], Xf[1], Xf[2]};
          double res[2], J[2][21];
          const int64_t ks = kset(cam);
          const double* kk = shared + 7 * C + 9 * ks;
          const uint32_t kmask = kmasks[(size_t)ks];
          rigk_eval(kk, Rf, &ft[f * 3], &Rc[cam * 9], &shared[cam * 7 + 4], X, ouv[o * 2], ouv[o * 2 + 1], res,
                    B ? J : nullptr);
          double rho = res[0] * res[0] + res[1] * res[1], sr = 1.0;
          if (huber_a > 0.0) huber(huber_a, rho, &rho, &sr);
          c += 0.5 * rho;
          if (obs_cost) obs_cost[o] = 0.5 * rho;
          if (!B) continue;
          const bool fixed = cam_fixed[cam];
          double* hss = hs[tid].data();
          for (int r = 0; r < 2; ++r) {
            // the row restricted to the shared block: 6 camera columns + 9 intrinsics columns
            double vs[15], vf[6];
            int col[15];
            for (int i = 0; i < 6; ++i) { vs[i] = fixed ? 0.0 : sr * J[r][i]; col[i] = (int)cam * 6 + i; vf[i] = sr * J[r][6 + i]; }
            for (int i = 0; i < 9; ++i) { vs[6 + i] = (kmask & (1u << i)) ? 0.0 : sr * J[r][12 + i]; col[6 + i] = K0 + (int)(9 * ks) + i; }
            const double rr = sr * res[r];
            for (int i = 0; i < 6; ++i) {
              for (int j = 0; j < 6; ++j) B->Hpp[f * 36 + i * 6 + j] += vf[i] * vf[j];
              for (int j = 0; j < 15; ++j) B->Hps[(f * 6 + i) * Sd + col[j]] += vf[i] * vs[j];
              B->gp[f * 6 + i] += vf[i] * rr;
            }
            for (int i = 0; i < 15; ++i) {
              for (int j = 0; j < 15; ++j) hss[(size_t)col[i] * Sd + col[j]] += vs[i] * vs[j];
              hss[(size_t)Sd * Sd + col[i]] += vs[i] * rr;
            }
          }
        }
      }
      cost_t[tid] = c;
    });
    double cost = 0;
    for (int t = 0; t < nt; ++t) {
      cost += cost_t[t];
      if (B) {
        for (size_t i = 0; i < (size_t)Sd * Sd; ++i) B->Hss[i] += hs[t][i];
        for (int i = 0; i < Sd; ++i) B->gs[i] += hs[t][(size_t)Sd * Sd + i];
      }
    }
    return cost;
  }
  void plus_shared(const double* shared, const double* delta, double* out) override {
    for (int64_t c = 0; c < C; ++c) {
      if (cam_fixed[c]) {
        for (int i = 0; i < 7; ++i) out[c * 7 + i] = shared[c * 7 + i];
        continue;
      }
      quat_plus(&shared[c * 7], &delta[c * 6], &out[c * 7]);
      for (int i = 0; i < 3; ++i) out[c * 7 + 4 + i] = shared[c * 7 + 4 + i] + delta[c * 6 + 3 + i];
    }
    for (int64_t ks = 0; ks < NK; ++ks)
      for (int i = 0; i < 9; ++i)
        out[7 * C + 9 * ks + i] = shared[7 * C + 9 * ks + i] + ((shared_fixed[6 * C + 9 * ks + i]) ? 0.0 : delta[6 * C + 9 * ks + i]);
  }
};

// ------------------------------------------------------------------------------------------
// float helpers restating Eigen-typed code of the reference
// ------------------------------------------------------------------------------------------

// 3x3 float inverse via cofactors (Eigen's fixed-size 3x3 inverse), row-major.
void inv3f(const float* m, float* o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.0f / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
inline void mat3vecf(const float* m, const float* v, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = m[i * 3] * v[0] + m[i * 3 + 1] * v[1] + m[i * 3 + 2] * v[2];
}
inline void normalize3f(float* v) {
  const float z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (z > 0) { const float s = std::sqrt(z); v[0] /= s; v[1] /= s; v[2] /= s; }
}
inline void cross3f(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// One-sided (Hestenes) Jacobi SVD of a row-major m x n matrix (m >= 1, n <= 16): returns V
// (n x n row-major, columns = right singular vectors) and sigma (column norms of A V).
// Stands in for Eigen::JacobiSVD at geometry.cpp:99,154,200.
void jacobi_svd(std::vector<double>& A, int m, int n, std::vector<double>& V, std::vector<double>& sig) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < m; ++i) {
          const double ap = A[(size_t)i * n + p], aq = A[(size_t)i * n + q];
          al += ap * ap; be += aq * aq; ga += ap * aq;
        }
        if (ga == 0.0 || std::fabs(ga) <= 1e-300 + 2.3e-16 * std::sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + tt * tt), s = c * tt;
        for (int i = 0; i < m; ++i) {
          const double ap = A[(size_t)i * n + p], aq = A[(size_t)i * n + q];
          A[(size_t)i * n + p] = c * ap - s * aq;
          A[(size_t)i * n + q] = s * ap + c * aq;
        }
        for (int i = 0; i < n; ++i) {
          const double vp = V[(size_t)i * n + p], vq = V[(size_t)i * n + q];
          V[(size_t)i * n + p] = c * vp - s * vq;
          V[(size_t)i * n + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  sig.assign(n, 0.0);
  for (int j = 0; j < n; ++j) {
    double a = 0;
    for (int i = 0; i < m; ++i) a += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    sig[j] = std::sqrt(a);
  }
}

// U V^T of the SVD of a 3x3 (double, row-major) = closest orthogonal matrix.
void polar3(const double* M, double* out) {
  std::vector<double> A(M, M + 9), V, sig;
  jacobi_svd(A, 3, 3, V, sig);
  // A now = U Sigma; out = U V^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += (A[i * 3 + k] / sig[k]) * V[j * 3 + k];
      out[i * 3 + j] = a;
    }
}

// Eigen::Quaternion(rotation matrix) (QuaternionBase::operator=(MatrixBase), Shepperd), w x y z
void mat_to_quat(const double* m, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (m[7] - m[5]) * t; q[2] = (m[2] - m[6]) * t; q[3] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[1 + j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[1 + k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

void oc_options_init(oc_options* o) {
  o->max_iterations = 100;
  o->use_nonmonotonic_steps = 1;
  o->max_consecutive_nonmonotonic_steps = 5;
  o->jacobi_scaling = 1;
  o->max_consecutive_invalid_steps = 5;
  o->num_threads = 1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
}

void oc_intrinsics_residual(const double* intr, const double* q, const double* t, const double* X,
                            const double* uv, double* res, double* J) {
  double R[9];
  quat_to_R(q, R);
  intr_eval(intr, R, t, X, uv[0], uv[1], res, reinterpret_cast<double(*)[15]>(J));
}

double oc_intrinsics_blocks(int64_t F, const int64_t* off, const float* uv, const float* xyz,
                            const double* intr, uint32_t mask, const double* q, const double* t,
                            double* blocks, int32_t num_threads) {
  const int nt = std::max(1, (int)num_threads);
  std::vector<double> cost_t(nt, 0.0);
  parallel_frames(F, nt, [&](int64_t f0, int64_t f1, int tid) {
    double G[256], c = 0;
    for (int64_t f = f0; f < f1; ++f) {
      const int64_t s = off[f], n = off[f + 1] - off[f];
      c += intr_frame_block(intr, mask, &q[f * 4], &t[f * 3], uv + s * 2, xyz + s * 3, n, G);
      if (blocks) std::memcpy(blocks + f * 256, G, sizeof(G));
    }
    cost_t[tid] = c;
  });
  double cost = 0;
  for (double c : cost_t) cost += c;
  return cost;
}

int oc_intrinsics_solve(const oc_options* opt, int64_t F, const int64_t* off, const float* uv,
                        const float* xyz, double* intr, uint32_t mask, double* q, double* t,
                        oc_summary* summary) {
  return intrinsics_solve_impl(opt, F, off, uv, xyz, intr, mask, q, t, summary, nullptr, nullptr);
}

int oc_intrinsics_solve_sharded(const oc_options* opt, int64_t F, const int64_t* off,
                                const float* uv, const float* xyz, double* intr, uint32_t mask,
                                double* q, double* t, oc_summary* summary, oc_allreduce_fn ar,
                                void* ctx) {
  return intrinsics_solve_impl(opt, F, off, uv, xyz, intr, mask, q, t, summary, ar, ctx);
}

// Calibrator::Distort (calibrator.cpp:157-166): DistortPixels<float,float>; the double literals in
// DistortNormalized (calibrator.cpp:80-82) promote the intermediate sums to double.
void oc_distort(const float* K, const float* d, int64_t n, const float* xy, float* out) {
  const float fx = K[0], fy = K[4], px = K[2], py = K[5];
  const float k1 = d[0], k2 = d[1], k3 = d[4], p1 = d[2], p2 = d[3];  // arg shuffle at :162
  for (int64_t i = 0; i < n; ++i) {
    const float x = xy[i * 2], y = xy[i * 2 + 1];
    const float r2 = x * x + y * y;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const double r_mult = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const float nx = (float)(x * r_mult + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x));
    const float ny = (float)(y * r_mult + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y));
    out[i * 2] = fx * nx + px;
    out[i * 2 + 1] = fy * ny + py;
  }
}

// Calibrator::Undistort (calibrator.cpp:118-155): per point, minimise |distort(x,y) - (xd,yd)|^2
// over (x,y) starting at (xd,yd).  The reference runs a 2-variable Ceres solve (DENSE_QR, <=1000
// iterations); restated as Levenberg-Marquardt with the same radius rules, iterated to the root.
void oc_undistort(const float* K, const float* d, int64_t n, const float* uv, float* out) {
  float Ki[9];
  inv3f(K, Ki);
  const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
  for (int64_t i = 0; i < n; ++i) {
    const float pxf = uv[i * 2], pyf = uv[i * 2 + 1];
    const float w = Ki[6] * pxf + Ki[7] * pyf + Ki[8];
    const double xd = (Ki[0] * pxf + Ki[1] * pyf + Ki[2]) / w;  // float expression, :137
    const double yd = (Ki[3] * pxf + Ki[4] * pyf + Ki[5]) / w;
    double x = xd, y = yd, radius = 1e4, dec = 2.0;
    auto f = [&](double x_, double y_, double* r, double* Jm) {
      const double r2 = x_ * x_ + y_ * y_, r4 = r2 * r2, r6 = r4 * r2;
      const double m = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
      r[0] = x_ * m + 2.0 * p1 * x_ * y_ + p2 * (r2 + 2.0 * x_ * x_) - xd;
      r[1] = y_ * m + 2.0 * p2 * x_ * y_ + p1 * (r2 + 2.0 * y_ * y_) - yd;
      if (Jm) {
        const double mp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
        Jm[0] = m + 2.0 * mp * x_ * x_ + 2.0 * p1 * y_ + 6.0 * p2 * x_;
        Jm[1] = Jm[2] = 2.0 * mp * x_ * y_ + 2.0 * p1 * x_ + 2.0 * p2 * y_;
        Jm[3] = m + 2.0 * mp * y_ * y_ + 2.0 * p2 * x_ + 6.0 * p1 * y_;
      }
    };
    double r[2], Jm[4];
    f(x, y, r, Jm);
    double cost = 0.5 * (r[0] * r[0] + r[1] * r[1]);
    for (int it = 0; it < 1000 && cost > 0.0; ++it) {
      const double h00 = Jm[0] * Jm[0] + Jm[2] * Jm[2], h01 = Jm[0] * Jm[1] + Jm[2] * Jm[3];
      const double h11 = Jm[1] * Jm[1] + Jm[3] * Jm[3];
      const double g0 = Jm[0] * r[0] + Jm[2] * r[1], g1 = Jm[1] * r[0] + Jm[3] * r[1];
      const double a00 = h00 + std::max(h00, 1e-6) / radius, a11 = h11 + std::max(h11, 1e-6) / radius;
      const double det = a00 * a11 - h01 * h01;
      const double dx = -(a11 * g0 - h01 * g1) / det, dy = -(a00 * g1 - h01 * g0) / det;
      if (!(std::fabs(dx) + std::fabs(dy) > 1e-17 * (std::fabs(x) + std::fabs(y) + 1e-300))) break;
      double rn[2], Jn[4];
      f(x + dx, y + dy, rn, Jn);
      const double cn = 0.5 * (rn[0] * rn[0] + rn[1] * rn[1]);
      if (cn < cost) {
        x += dx; y += dy; cost = cn;
        r[0] = rn[0]; r[1] = rn[1];
        std::copy(Jn, Jn + 4, Jm);
        radius = std::min(1e16, radius * 3.0);
        dec = 2.0;
      } else {
        radius /= dec; dec *= 2.0;
        if (radius < 1e-32) break;
      }
    }
    out[i * 2] = (float)x;
    out[i * 2 + 1] = (float)y;
  }
}

void oc_rig_residual(const double* q_rw, const double* t_rw, const double* q_cr, const double* t_cr,
                     const double* X, const double* uv, double* res, double* J) {
  double Rf[9], Rc[9];
  quat_to_R(q_rw, Rf);
  quat_to_R(q_cr, Rc);
  rig_eval(Rf, t_rw, Rc, t_cr, X, uv[0], uv[1], res, reinterpret_cast<double(*)[12]>(J));
}

static int rig_solve_impl(const oc_options* opt, int64_t C, int64_t F, const int64_t* off,
                          const uint32_t* ocam, const uint64_t* oworld, const float* ouv, const float* wxyz,
                          double* cam_q, double* cam_t, const uint8_t* cam_frozen, const uint8_t* cam_seen_global,
                          double* frame_q, double* frame_t, double huber_a, double* obs_cost, oc_summary* summary,
                          oc_allreduce_fn ar, void* ctx) {
  oc_options o;
  if (opt) o = *opt; else { oc_options_init(&o); o.max_iterations = 1000; }
  RigProblem P;
  P.C = C; P.F = F; P.S = (int)(6 * C); P.S_amb = (int)(7 * C);
  P.shared_pose_blocks = C;
  P.off = off; P.ocam = ocam; P.oworld = oworld; P.ouv = ouv; P.wxyz = wxyz;
  P.huber_a = huber_a;
  P.num_threads = o.num_threads;
  // cameras / frames with no observation never enter the problem (extrinsics_calibrator.cpp:155-204)
  std::vector<uint8_t> cam_seen(C, 0);
  P.frame_active.assign(F, 0);
  for (int64_t f = 0; f < F; ++f)
    for (int64_t k = off[f]; k < off[f + 1]; ++k) { cam_seen[ocam[k]] = 1; P.frame_active[f] = 1; }
  if (cam_seen_global) for (int64_t c = 0; c < C; ++c) cam_seen[c] = cam_seen_global[c];
  P.cam_fixed.resize(C);
  P.shared_fixed.resize(6 * C);
  P.shared_amb_active.resize(7 * C);
  for (int64_t c = 0; c < C; ++c) {
    P.cam_fixed[c] = (cam_frozen && cam_frozen[c]) || !cam_seen[c];
    for (int i = 0; i < 6; ++i) P.shared_fixed[c * 6 + i] = P.cam_fixed[c];
    for (int i = 0; i < 7; ++i) P.shared_amb_active[c * 7 + i] = !P.cam_fixed[c];
  }
  std::vector<double> shared(7 * C);
  for (int64_t c = 0; c < C; ++c) {
    for (int i = 0; i < 4; ++i) shared[c * 7 + i] = cam_q[c * 4 + i];
    for (int i = 0; i < 3; ++i) shared[c * 7 + 4 + i] = cam_t[c * 3 + i];
  }
  const int rc = run_lm(P, o, shared.data(), frame_q, frame_t, summary, ar, ctx);
  for (int64_t c = 0; c < C; ++c) {
    for (int i = 0; i < 4; ++i) cam_q[c * 4 + i] = shared[c * 7 + i];
    for (int i = 0; i < 3; ++i) cam_t[c * 3 + i] = shared[c * 7 + 4 + i];
  }
  if (obs_cost) {  // extrinsics_calibrator.cpp:219-225
    P.obs_cost = obs_cost;
    P.eval(shared.data(), frame_q, frame_t, nullptr);
  }
  return rc;
}

int oc_rig_solve(const oc_options* opt, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                 const uint32_t* ocam, const uint64_t* oworld, const float* ouv, const float* wxyz,
                 double* cam_q, double* cam_t, const uint8_t* cam_frozen, double* frame_q,
                 double* frame_t, double huber_a, double* obs_cost, oc_summary* summary) {
  (void)n_world;
  return rig_solve_impl(opt, C, F, off, ocam, oworld, ouv, wxyz, cam_q, cam_t, cam_frozen, nullptr, frame_q, frame_t,
                        huber_a, obs_cost, summary, nullptr, nullptr);
}

int oc_rig_solve_sharded(const oc_options* opt, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                         const uint32_t* ocam, const uint64_t* oworld, const float* ouv, const float* wxyz,
                         double* cam_q, double* cam_t, const uint8_t* cam_frozen, const uint8_t* cam_seen_global,
                         double* frame_q, double* frame_t, double huber_a, double* obs_cost, oc_summary* summary,
                         oc_allreduce_fn ar, void* ctx) {
  (void)n_world;
  return rig_solve_impl(opt, C, F, off, ocam, oworld, ouv, wxyz, cam_q, cam_t, cam_frozen, cam_seen_global, frame_q,
                        frame_t, huber_a, obs_cost, summary, ar, ctx);
}

// EXTENSION: rig poses + shared intrinsics on pixel observations (see RigKProblem)
void oc_rigk_residual(const double* k, const double* q_rw, const double* t_rw, const double* q_cr, const double* t_cr,
                      const double* X, const double* uv, double* res, double* J) {
  double Rf[9], Rc[9];
  quat_to_R(q_rw, Rf);
  quat_to_R(q_cr, Rc);
  rigk_eval(k, Rf, t_rw, Rc, t_cr, X, uv[0], uv[1], res, reinterpret_cast<double(*)[21]>(J));
}

// per_camera == 0: intr[9], kmask[1] (one set shared by all cameras); != 0: intr[C][9], kmask[C]. A set that no
// observation uses is not part of the problem (its coordinates stay put and do not count in |x|).
int oc_rigk_solve_sets(const oc_options* opt, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                       const uint32_t* ocam, const uint64_t* oworld, const float* ouv, const float* wxyz,
                       int32_t per_camera, double* intr, const uint32_t* kmask, double* cam_q, double* cam_t,
                       const uint8_t* cam_frozen, double* frame_q, double* frame_t, double huber_a, double* obs_cost,
                       oc_summary* summary) {
  (void)n_world;
  oc_options o;
  if (opt) o = *opt; else { oc_options_init(&o); o.max_iterations = 1000; }
  RigKProblem P;
  const int64_t NK = per_camera ? C : 1;
  P.NK = NK;
  P.C = C; P.F = F; P.S = (int)(6 * C + 9 * NK); P.S_amb = (int)(7 * C + 9 * NK);
  P.shared_pose_blocks = C;
  P.off = off; P.ocam = ocam; P.oworld = oworld; P.ouv = ouv; P.wxyz = wxyz;
  P.huber_a = huber_a;
  P.kmasks.assign(kmask, kmask + NK);
  P.num_threads = o.num_threads;
  std::vector<uint8_t> cam_seen(C, 0);
  P.frame_active.assign(F, 0);
  for (int64_t f = 0; f < F; ++f)
    for (int64_t k = off[f]; k < off[f + 1]; ++k) { cam_seen[ocam[k]] = 1; P.frame_active[f] = 1; }
  P.cam_fixed.resize(C);
  P.shared_fixed.assign(P.S, 0);
  P.shared_amb_active.assign(P.S_amb, 1);
  for (int64_t c = 0; c < C; ++c) {
    P.cam_fixed[c] = (cam_frozen && cam_frozen[c]) || !cam_seen[c];
    for (int i = 0; i < 6; ++i) P.shared_fixed[c * 6 + i] = P.cam_fixed[c];
    for (int i = 0; i < 7; ++i) P.shared_amb_active[c * 7 + i] = !P.cam_fixed[c];
  }
  for (int64_t ks = 0; ks < NK; ++ks) {
    const bool used = per_camera ? cam_seen[ks] != 0 : true;
    for (int i = 0; i < 9; ++i) {
      P.shared_fixed[6 * C + 9 * ks + i] = !used || ((kmask[ks] >> i) & 1);
      P.shared_amb_active[7 * C + 9 * ks + i] = used;
    }
  }
  std::vector<double> shared(P.S_amb);
  for (int64_t c = 0; c < C; ++c) {
    for (int i = 0; i < 4; ++i) shared[c * 7 + i] = cam_q[c * 4 + i];
    for (int i = 0; i < 3; ++i) shared[c * 7 + 4 + i] = cam_t[c * 3 + i];
  }
  for (int64_t i = 0; i < 9 * NK; ++i) shared[7 * C + i] = intr[i];
  const int rc = run_lm(P, o, shared.data(), frame_q, frame_t, summary, nullptr, nullptr);
  for (int64_t c = 0; c < C; ++c) {
    for (int i = 0; i < 4; ++i) cam_q[c * 4 + i] = shared[c * 7 + i];
    for (int i = 0; i < 3; ++i) cam_t[c * 3 + i] = shared[c * 7 + 4 + i];
  }
  for (int64_t i = 0; i < 9 * NK; ++i) intr[i] = shared[7 * C + i];
  if (obs_cost) {
    P.obs_cost = obs_cost;
    P.eval(shared.data(), frame_q, frame_t, nullptr);
  }
  return rc;
}

int oc_rigk_solve(const oc_options* opt, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                  const uint32_t* ocam, const uint64_t* oworld, const float* ouv, const float* wxyz,
                  double* intr, uint32_t kmask, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                  double* frame_q, double* frame_t, double huber_a, double* obs_cost, oc_summary* summary) {
  return oc_rigk_solve_sets(opt, C, F, n_world, off, ocam, oworld, ouv, wxyz, 0, intr, &kmask, cam_q, cam_t, cam_frozen,
                            frame_q, frame_t, huber_a, obs_cost, summary);
}

// ---- Zhang initialisation ----------------------------------------------------------------

// EstimateHomography (geometry.cpp:70-105): DLT, A is 2n x 9 double, h = last right singular
// vector. p1/p2 are float arrays with the given stride (2 or 3); only x,y are used.
void oc_estimate_homography(int64_t n, const float* p1, int32_t s1, const float* p2, int32_t s2,
                            float* H) {
  std::vector<double> A((size_t)2 * n * 9, 0.0), V, sig;
  for (int64_t i = 0; i < n; ++i) {
    const float x1 = p1[i * s1], y1 = p1[i * s1 + 1], x2 = p2[i * s2], y2 = p2[i * s2 + 1];
    double* r0 = &A[(size_t)(2 * i) * 9];
    double* r1 = r0 + 9;
    r0[3] = -x1; r0[4] = -y1; r0[5] = -1.0f;
    r0[6] = x1 * y2; r0[7] = y1 * y2; r0[8] = y2;   // float products, as geometry.cpp:86-88
    r1[0] = x1; r1[1] = y1; r1[2] = 1.0f;
    r1[6] = -x1 * x2; r1[7] = -y1 * x2; r1[8] = -x2;
  }
  jacobi_svd(A, (int)(2 * n), 9, V, sig);
  int m = 0;
  for (int j = 1; j < 9; ++j) if (sig[j] < sig[m]) m = j;
  for (int i = 0; i < 9; ++i) H[i] = (float)V[(size_t)i * 9 + m];
}

// EstimateKFromHomographies (geometry.cpp:123-177). Hs: n x 9 row-major floats.
void oc_estimate_k_from_homographies(int64_t n, const float* Hs, float* K) {
  std::vector<double> A((size_t)(2 * n + 1) * 6, 0.0), V, sig;
  auto vij = [](const double* H, int i, int j, double* v) {
    // H.col(i)(r) = H[r*3+i]
    v[0] = H[0 + i] * H[0 + j];
    v[1] = H[0 + i] * H[3 + j] + H[3 + i] * H[0 + j];
    v[2] = H[3 + i] * H[3 + j];
    v[3] = H[6 + i] * H[0 + j] + H[0 + i] * H[6 + j];
    v[4] = H[6 + i] * H[3 + j] + H[3 + i] * H[6 + j];
    v[5] = H[6 + i] * H[6 + j];
  };
  for (int64_t i = 0; i < n; ++i) {
    double H[9], a[6], b[6];
    for (int k = 0; k < 9; ++k) H[k] = Hs[i * 9 + k];
    vij(H, 0, 1, &A[(size_t)(2 * i) * 6]);
    vij(H, 0, 0, a);
    vij(H, 1, 1, b);
    for (int k = 0; k < 6; ++k) A[(size_t)(2 * i + 1) * 6 + k] = a[k] - b[k];
  }
  A[(size_t)(2 * n) * 6 + 1] = (double)n;  // zero-skew row weighted by n (geometry.cpp:150-152)
  jacobi_svd(A, (int)(2 * n + 1), 6, V, sig);
  int m = 0;
  for (int j = 1; j < 6; ++j) if (sig[j] < sig[m]) m = j;
  const double B11 = V[0 * 6 + m], B12 = V[1 * 6 + m], B22 = V[2 * 6 + m], B13 = V[3 * 6 + m],
               B23 = V[4 * 6 + m], B33 = V[5 * 6 + m];
  const double v0 = (B12 * B13 - B11 * B23) / (B11 * B22 - B12 * B12);
  const double l = B33 - (B13 * B13 + v0 * (B12 * B13 - B11 * B23)) / B11;
  const double alpha = std::sqrt(l / B11);
  const double beta = std::sqrt(l * B11 / (B11 * B22 - B12 * B12));
  const float y = 0.0f;
  const double u0 = y * v0 / beta - B13 * alpha * alpha / l;
  K[0] = (float)alpha; K[1] = y; K[2] = (float)u0;
  K[3] = 0.0f; K[4] = (float)beta; K[5] = (float)v0;
  K[6] = 0.0f; K[7] = 0.0f; K[8] = 1.0f;
}

// FixRotationMatrix (geometry.cpp:197-203): U V^T of the SVD.
void oc_fix_rotation_matrix(const float* R, float* out) {
  double M[9], P[9];
  for (int i = 0; i < 9; ++i) M[i] = R[i];
  polar3(M, P);
  for (int i = 0; i < 9; ++i) out[i] = (float)P[i];
}

// RecoverExtrinsics (geometry.cpp:179-195), float arithmetic.
void oc_recover_extrinsics(const float* Ki, const float* H, float* Rout, float* t) {
  float h0[3] = {H[0], H[3], H[6]}, h1[3] = {H[1], H[4], H[7]}, h2[3] = {H[2], H[5], H[8]};
  float a0[3], a1[3], a2[3];
  mat3vecf(Ki, h0, a0);
  const float l = 1.0f / std::sqrt(a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]);
  float lK[9];
  for (int i = 0; i < 9; ++i) lK[i] = l * Ki[i];
  float r0[3] = {l * a0[0], l * a0[1], l * a0[2]}, r1[3], r2[3];
  mat3vecf(lK, h1, r1);
  cross3f(r0, r1, r2);
  mat3vecf(lK, h2, a2);
  (void)a1;
  float R[9] = {r0[0], r1[0], r2[0], r0[1], r1[1], r2[1], r0[2], r1[2], r2[2]};
  oc_fix_rotation_matrix(R, Rout);
  t[0] = a2[0]; t[1] = a2[1]; t[2] = a2[2];
}

// Calibrator::Estimate up to the Optimize call (calibrator.cpp:47-66).
// NOTE: the sign of the DLT null vector is arbitrary (Eigen's JacobiSVD fixes it one way, this
// code fixes it so that t_z > 0); H and -H give mirrored poses with identical projections, so
// the optimum of the following LM is unaffected.
void oc_zhang_init(int64_t F, const int64_t* off, const float* uv, const float* xyz, float* K,
                   float* q, float* t) {
  std::vector<float> Hs((size_t)F * 9);
  for (int64_t f = 0; f < F; ++f)
    oc_estimate_homography(off[f + 1] - off[f], xyz + off[f] * 3, 3, uv + off[f] * 2, 2, &Hs[f * 9]);
  oc_estimate_k_from_homographies(F, Hs.data(), K);
  float Ki[9];
  inv3f(K, Ki);
  for (int64_t f = 0; f < F; ++f) {
    float H[9];
    for (int i = 0; i < 9; ++i) H[i] = Hs[f * 9 + i];
    float R[9], tt[3];
    oc_recover_extrinsics(Ki, H, R, tt);
    if (tt[2] < 0) {
      for (int i = 0; i < 9; ++i) H[i] = -H[i];
      oc_recover_extrinsics(Ki, H, R, tt);
    }
    double Rd[9], qd[4];
    for (int i = 0; i < 9; ++i) Rd[i] = R[i];
    mat_to_quat(Rd, qd);  // qs.emplace_back(R) -> Quaternionf(Matrix3f), calibrator.cpp:63
    for (int i = 0; i < 4; ++i) q[f * 4 + i] = (float)qd[i];
    for (int i = 0; i < 3; ++i) t[f * 3 + i] = tt[i];
  }
}

// ---- synthetic data ------------------------------------------------------------------------

struct oc_generator {
  int width, height;
  float min_distance = 0.2f, max_distance = 1.0f, noise = 0.0f;  // data_generator.hh:38-40
  float K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float dist[5] = {0, 0, 0, 0, 0};
  std::mt19937 gen{0};  // data_generator.hh:43
  std::uniform_real_distribution<float> rand_w, rand_h, rand_dist, rand_pixel;
  oc_generator(int w, int h)
      : width(w), height(h), rand_w(0.0f, w - 1.0f), rand_h(0.0f, h - 1.0f),
        rand_dist(0.2f, 1.0f), rand_pixel(-0.0f, 0.0f) {}
};

oc_generator* oc_generator_create(int32_t w, int32_t h) { return new oc_generator(w, h); }
void oc_generator_destroy(oc_generator* g) { delete g; }
void oc_generator_set_k(oc_generator* g, const float* K) { std::memcpy(g->K, K, sizeof(g->K)); }
void oc_generator_set_distortion(oc_generator* g, const float* d) { std::memcpy(g->dist, d, sizeof(g->dist)); }
void oc_generator_set_noise(oc_generator* g, float noise) {
  g->noise = noise;  // data_generator.cpp:46-50
  g->rand_pixel = std::uniform_real_distribution<float>(-noise, noise);
}

namespace {
// ProjectToCameraAndDistort (data_generator.cpp:10-32): cv::projectPoints with rvec = tvec = 0
// (OpenCV's radial-tangential model evaluated in double, result stored as float), then noise.
bool project_and_distort(oc_generator* g, const float* p, float* uv) {
  const double x = (double)p[0] / (double)p[2], y = (double)p[1] / (double)p[2];
  const double k1 = g->dist[0], k2 = g->dist[1], p1 = g->dist[2], p2 = g->dist[3], k3 = g->dist[4];
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
  const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  const double xd = x * cdist + p1 * a1 + p2 * a2, yd = y * cdist + p1 * a3 + p2 * a1;
  const float u = (float)(xd * (double)g->K[0] + (double)g->K[2]);
  const float v = (float)(yd * (double)g->K[4] + (double)g->K[5]);
  const float ud = u + g->rand_pixel(g->gen);
  const float vd = v + g->rand_pixel(g->gen);
  if (ud < 0.0f || ud >= g->width - 1 || vd < 0.0f || vd >= g->height - 1) return false;
  uv[0] = ud; uv[1] = vd;
  return true;
}
// GetRandom3DPointVisibleToCamera (data_generator.cpp:126-146)
void random_visible_point(oc_generator* g, const float* Kinv, float* p3) {
  float p[3];
  p[0] = g->rand_w(g->gen);
  p[1] = g->rand_h(g->gen);
  p[2] = 1.0f;
  mat3vecf(Kinv, p, p3);
  normalize3f(p3);
  const float z = g->rand_dist(g->gen);
  p3[0] *= z; p3[1] *= z; p3[2] *= z;
}
}  // namespace

int64_t oc_generator_planar(oc_generator* g, int32_t num_p, float* uv, float* xyz) {
  float Kinv[9];
  inv3f(g->K, Kinv);  // pseudoInverse of an invertible K (data_generator.cpp:81)
  // GetRandomPlane (data_generator.cpp:52-75)
  float corner[3] = {0.0f, 0.0f, 1.0f}, P[3][3];
  mat3vecf(Kinv, corner, P[0]); normalize3f(P[0]);
  { const float s = g->rand_dist(g->gen); for (int i = 0; i < 3; ++i) P[0][i] *= s; }
  corner[0] = (float)(g->width - 1);
  mat3vecf(Kinv, corner, P[1]); normalize3f(P[1]);
  { const float s = g->rand_dist(g->gen); for (int i = 0; i < 3; ++i) P[1][i] *= s; }
  corner[1] = (float)(g->height - 1);
  mat3vecf(Kinv, corner, P[2]); normalize3f(P[2]);
  { const float s = g->rand_dist(g->gen); for (int i = 0; i < 3; ++i) P[2][i] *= s; }
  // EstimatePlaneFinite (geometry.cpp:6-17)
  float A[9] = {P[0][0], P[0][1], P[0][2], P[1][0], P[1][1], P[1][2], P[2][0], P[2][1], P[2][2]};
  float Ai[9], ones[3] = {1.0f, 1.0f, 1.0f}, plane[4];
  inv3f(A, Ai);
  mat3vecf(Ai, ones, plane);
  plane[3] = -1.0f;
  // RotationMatrixFromPlane (geometry.cpp:23-39), new_normal = UnitZ
  float normal[3] = {plane[0], plane[1], plane[2]}, zaxis[3] = {0, 0, 1}, v1[3], v2[3];
  normalize3f(normal);
  cross3f(normal, zaxis, v1); normalize3f(v1);
  cross3f(normal, v1, v2); normalize3f(v2);
  const float R[9] = {v1[0], v1[1], v1[2], v2[0], v2[1], v2[2], normal[0], normal[1], normal[2]};
  int64_t rejected = 0;
  int32_t have = 0;
  while (have < num_p) {
    float p3[3];
    random_visible_point(g, Kinv, p3);
    // ProjectToPlane(plane, p, p) -> double internally (geometry.cpp:41-68)
    const double pn[3] = {plane[0], plane[1], plane[2]};
    const double pd[3] = {p3[0], p3[1], p3[2]};
    const double dot = pd[0] * pn[0] + pd[1] * pn[1] + pd[2] * pn[2];
    const double tt = (dot + (double)plane[3]) / dot;
    const float planar[3] = {(float)(pd[0] - pd[0] * tt), (float)(pd[1] - pd[1] * tt), (float)(pd[2] - pd[2] * tt)};
    float rot[3];
    mat3vecf(R, planar, rot);
    float px[2];
    if (!project_and_distort(g, planar, px)) { ++rejected; continue; }
    xyz[have * 3] = rot[0]; xyz[have * 3 + 1] = rot[1]; xyz[have * 3 + 2] = 0.0f;  // :119
    uv[have * 2] = px[0]; uv[have * 2 + 1] = px[1];
    ++have;
  }
  return rejected;
}

int64_t oc_generator_points(oc_generator* g, int32_t num_p, float* uv, float* xyz) {
  float Kinv[9];
  inv3f(g->K, Kinv);
  int64_t rejected = 0;
  int32_t have = 0;
  while (have < num_p) {
    float p3[3], px[2];
    random_visible_point(g, Kinv, p3);
    if (!project_and_distort(g, p3, px)) { ++rejected; continue; }
    xyz[have * 3] = p3[0]; xyz[have * 3 + 1] = p3[1]; xyz[have * 3 + 2] = p3[2];
    uv[have * 2] = px[0]; uv[have * 2 + 1] = px[1];
    ++have;
  }
  return rejected;
}

// ---- rig scenario (test_extrinsics_calibrator.cpp:9-134) ------------------------------------
namespace {
struct Aff { float R[9]; float t[3]; };  // row-major linear part
void aff_identity(Aff& a) { const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; std::memcpy(a.R, I, sizeof(I)); a.t[0] = a.t[1] = a.t[2] = 0; }
void mat3mulf(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
void rot_axis(int axis, float ang, float* R) {
  const float c = std::cos(ang), s = std::sin(ang);
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(R, I, sizeof(I));
  if (axis == 2) { R[0] = c; R[1] = -s; R[3] = s; R[4] = c; }
  else if (axis == 1) { R[0] = c; R[2] = s; R[6] = -s; R[8] = c; }
  else { R[4] = c; R[5] = -s; R[7] = s; R[8] = c; }
}
// DistortTransformation (test_extrinsics_calibrator.cpp:9-38). The three angle draws appear in one
// C++ expression whose operand evaluation order is unspecified; drawn here left to right.
Aff distort_transformation(const Aff& T, std::mt19937& gen, float terr, float rerr_deg) {
  std::uniform_real_distribution<float> rt(-terr, terr), rr(-rerr_deg, rerr_deg);
  const float pi = 3.141592653589793f;
  float Rz1[9], Ry[9], Rz2[9], tmp[9], Rerr[9];
  rot_axis(2, rr(gen) / 180.0f * pi, Rz1);
  rot_axis(1, rr(gen) / 180.0f * pi, Ry);
  rot_axis(2, rr(gen) / 180.0f * pi, Rz2);
  mat3mulf(Rz1, Ry, tmp);
  mat3mulf(tmp, Rz2, Rerr);
  Aff o;
  mat3mulf(T.R, Rerr, o.R);
  for (int i = 0; i < 3; ++i) o.t[i] = T.t[i] + rt(gen);
  return o;
}
void aff_store(const Aff& a, float* T16) {  // column-major 4x4 (extrinsics_calibrator.cpp:271)
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) T16[c * 4 + r] = a.R[r * 3 + c];
    T16[c * 4 + 3] = 0.0f;
  }
  T16[12] = a.t[0]; T16[13] = a.t[1]; T16[14] = a.t[2]; T16[15] = 1.0f;
}
void aff_apply(const Aff& a, const float* p, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = a.R[i * 3] * p[0] + a.R[i * 3 + 1] * p[1] + a.R[i * 3 + 2] * p[2] + a.t[i];
}
}  // namespace

void oc_rig_scenario(int32_t C, int32_t F, int32_t M, uint32_t seed, float* cam_T, float* cam_T_true,
                     float* frame_T, float* world_xyz, uint32_t* obs_cam, uint64_t* obs_world,
                     float* obs_uv) {
  std::mt19937 gen{seed};
  std::uniform_real_distribution<float> rand_trans_rig(-0.03f, 0.03f);
  std::vector<Aff> cams(C), cams_d(C);
  aff_identity(cams[0]);
  for (int i = 1; i < C; ++i) {
    aff_identity(cams[i]);
    cams[i].t[0] = rand_trans_rig(gen);
    cams[i].t[1] = rand_trans_rig(gen);
  }
  for (int i = 0; i < C; ++i) {
    cams_d[i] = i == 0 ? cams[i] : distort_transformation(cams[i], gen, 0.005f, 0.1f);
    aff_store(cams_d[i], cam_T + i * 16);
    aff_store(cams[i], cam_T_true + i * 16);
  }
  std::uniform_real_distribution<float> rand_trans(0.3f, 1.0f), rand_pt(-0.2f, 0.2f);
  const float e2 = 2.0f / 500.0f, e3 = 0.001f;
  std::uniform_real_distribution<float> err2(-e2, e2), err3(-e3, e3);
  int64_t wp = 0, ob = 0;
  for (int f = 0; f < F; ++f) {
    Aff T;
    T.t[0] = rand_trans(gen); T.t[1] = rand_trans(gen); T.t[2] = rand_trans(gen);
    float fw[3] = {T.t[0], T.t[1], T.t[2]}, yax[3] = {0, 1, 0}, right[3], up[3];
    normalize3f(fw);
    cross3f(yax, fw, right); normalize3f(right);
    cross3f(fw, right, up);
    for (int i = 0; i < 3; ++i) { T.R[i] = fw[i]; T.R[3 + i] = right[i]; T.R[6 + i] = up[i]; }
    const Aff Td = distort_transformation(T, gen, 0.02f, 1.0f);
    aff_store(Td, frame_T + f * 16);
    for (int p = 0; p < M; ++p) {
      float X[3] = {rand_pt(gen), rand_pt(gen), rand_pt(gen)};
      float Xd[3] = {X[0], X[1], X[2]};
      Xd[0] += err3(gen); Xd[1] += err3(gen); Xd[2] += err3(gen);
      for (int i = 0; i < 3; ++i) world_xyz[wp * 3 + i] = Xd[i];
      for (int c = 0; c < C; ++c) {
        float xr[3], xc[3];
        aff_apply(T, X, xr);
        aff_apply(cams[c], xr, xc);
        float u = xc[0] / xc[2], v = xc[1] / xc[2];
        u += err2(gen);
        v += err2(gen);
        obs_cam[ob] = (uint32_t)c; obs_world[ob] = (uint64_t)wp;
        obs_uv[ob * 2] = u; obs_uv[ob * 2 + 1] = v;
        ++ob;
      }
      ++wp;
    }
  }
}

// Affine3f -> Quaterniond(rotation().cast<double>()) + translation (extrinsics_calibrator.cpp:116-130).
// Eigen's Transform::rotation() extracts the closest rotation via SVD; done here in double.
void oc_affine_to_qt(const float* T16, double* q, double* t) {
  double M[9], R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M[r * 3 + c] = T16[c * 4 + r];
  polar3(M, R);
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0) for (int i = 0; i < 9; ++i) R[i] = -R[i];
  mat_to_quat(R, q);
  t[0] = T16[12]; t[1] = T16[13]; t[2] = T16[14];
}

// extrinsics_calibrator.cpp:228-256: q components cast to float, Quaterniond, normalized(),
// toRotationMatrix(), cast<float>; translation cast to float.
void oc_qt_to_affine(const double* qd, const double* t, float* T16) {
  double w = (float)qd[0], x = (float)qd[1], y = (float)qd[2], z = (float)qd[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);  // Eigen coeffs order x y z w
  w /= n; x /= n; y /= n; z /= n;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz),
                       tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) T16[c * 4 + r] = (float)R[r * 3 + c];
    T16[c * 4 + 3] = 0.0f;
  }
  T16[12] = (float)t[0]; T16[13] = (float)t[1]; T16[14] = (float)t[2]; T16[15] = 1.0f;
}

}  // extern "C"
