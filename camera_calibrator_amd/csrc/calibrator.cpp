// calibrator.cpp -- Calibrator on top of the C ABI (include/cc_solver.h).
#include "calibrator.hh"

#include <cassert>
#include <cstdint>
#include <stdexcept>

#include "../../include/cc_solver.h"
#include "geometry.hh"

namespace calibrator {

namespace {
// parameter order of the shared block, identical to the reference's enum (calibrator.cpp:168-179)
enum Intrinsic { FX, FY, PX, PY, K1, K2, P1, P2, K3, kNumIntrinsics };
}  // namespace

Calibrator::Calibrator(const int img_width, const int img_height) : width_(img_width), height_(img_height) {}

void Calibrator::EstimateOpenCv(const std::vector<Points2D>&, const std::vector<Points3D>&) {
  (void)width_;
  (void)height_;
  throw std::runtime_error("Calibrator::EstimateOpenCv wraps cv::calibrateCamera; OpenCV is not part of the MI355X build");
}

void Calibrator::Estimate(const std::vector<Points2D>& in_img_points, const std::vector<Points3D>& in_world_points) {
  assert(in_img_points.size() == in_world_points.size());
  const size_t n_img = in_img_points.size();
  // Zhang initialisation on the device (cc_zhang_init): homographies -> K -> poses
  std::vector<int64_t> offsets(n_img + 1, 0);
  for (size_t i = 0; i < n_img; ++i) offsets[i + 1] = offsets[i] + (int64_t)in_img_points[i].size();
  std::vector<float> uv((size_t)offsets[n_img] * 2), xyz((size_t)offsets[n_img] * 3);
  for (size_t i = 0; i < n_img; ++i) {
    size_t k = (size_t)offsets[i];
    for (size_t j = 0; j < in_img_points[i].size(); ++j, ++k) {
      uv[2 * k] = in_img_points[i][j].x(); uv[2 * k + 1] = in_img_points[i][j].y();
      xyz[3 * k] = in_world_points[i][j].x(); xyz[3 * k + 1] = in_world_points[i][j].y(); xyz[3 * k + 2] = in_world_points[i][j].z();
    }
  }
  float K9[9];
  std::vector<float> q(4 * n_img), t(3 * n_img);
  const int rc = cc_zhang_init(device_, (int64_t)n_img, offsets.data(), uv.data(), xyz.data(), K9, q.data(), t.data(), nullptr);
  if (rc != 0) throw std::runtime_error(std::string("Calibrator::Estimate: ") + cc_last_error());
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K_(r, c) = K9[r * 3 + c];
  std::vector<Quaternion> qs;
  std::vector<Point3D> ts;
  for (size_t i = 0; i < n_img; ++i) {
    qs.emplace_back(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);
    ts.emplace_back(t[3 * i], t[3 * i + 1], t[3 * i + 2]);
  }
  Optimize(in_img_points, in_world_points, qs, ts);
}

void Calibrator::Optimize(const std::vector<Points2D>& in_img_points, const std::vector<Points3D>& in_world_points,
                          std::vector<Quaternion>& qs, std::vector<Point3D>& ts) {
  const size_t n_img = in_img_points.size();
  assert(n_img == in_world_points.size() && n_img == qs.size() && n_img == ts.size());
  // CSR layout of the ragged frames + fp64 parameter arrays
  std::vector<int64_t> offsets(n_img + 1, 0);
  for (size_t i = 0; i < n_img; ++i) {
    assert(in_img_points[i].size() == in_world_points[i].size());
    offsets[i + 1] = offsets[i] + (int64_t)in_img_points[i].size();
  }
  std::vector<float> uv((size_t)offsets[n_img] * 2), xyz((size_t)offsets[n_img] * 3);
  std::vector<double> q(4 * n_img), t(3 * n_img);
  for (size_t i = 0; i < n_img; ++i) {
    size_t k = (size_t)offsets[i];
    for (size_t j = 0; j < in_img_points[i].size(); ++j, ++k) {
      uv[2 * k] = in_img_points[i][j].x();
      uv[2 * k + 1] = in_img_points[i][j].y();
      xyz[3 * k] = in_world_points[i][j].x();
      xyz[3 * k + 1] = in_world_points[i][j].y();
      xyz[3 * k + 2] = in_world_points[i][j].z();
    }
    q[4 * i] = qs[i].w(); q[4 * i + 1] = qs[i].x(); q[4 * i + 2] = qs[i].y(); q[4 * i + 3] = qs[i].z();
    t[3 * i] = ts[i].x(); t[3 * i + 1] = ts[i].y(); t[3 * i + 2] = ts[i].z();
  }
  double intr[kNumIntrinsics];
  intr[FX] = K_(0, 0); intr[FY] = K_(1, 1); intr[PX] = K_(0, 2); intr[PY] = K_(1, 2);
  intr[K1] = dist_(0); intr[K2] = dist_(1); intr[P1] = dist_(2); intr[P2] = dist_(3); intr[K3] = dist_(4);
  uint32_t frozen = 0;
  for (int idx : constant_intrinsics_)
    if (idx >= 0 && idx < kNumIntrinsics) frozen |= 1u << idx;

  cc_options options;
  cc_options_init(&options);  // non-monotonic steps, 100 iterations: calibrator.cpp:314-321
  cc_summary summary{};
  last_status_ = n_img == 0 ? 0
                            : cc_intrinsics_optimize(&options, device_, (int64_t)n_img, offsets.data(), uv.data(), xyz.data(),
                                                     intr, frozen, q.data(), t.data(), &summary);
  last_iterations_ = summary.iterations;
  last_final_cost_ = summary.final_cost;
  if (last_status_ == CC_ERR_NO_DEVICE || last_status_ == CC_ERR_HIP)
    throw std::runtime_error(std::string("Calibrator::Optimize: ") + cc_last_error());  // no silent CPU path

  K_(0, 0) = static_cast<float>(intr[FX]);
  K_(1, 1) = static_cast<float>(intr[FY]);
  K_(0, 2) = static_cast<float>(intr[PX]);
  K_(1, 2) = static_cast<float>(intr[PY]);
  dist_(0) = static_cast<float>(intr[K1]);
  dist_(1) = static_cast<float>(intr[K2]);
  dist_(2) = static_cast<float>(intr[P1]);
  dist_(3) = static_cast<float>(intr[P2]);
  dist_(4) = static_cast<float>(intr[K3]);
}

void Calibrator::ForceDistortionToConstant(const int distortion_idx) {
  constant_intrinsics_.insert(distortion_idx + static_cast<int>(K1));
}

namespace {
void flatten(const Matrix3& K, const DynamicVector& d, float* K9, float* d5) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K9[r * 3 + c] = K(r, c);
  for (int i = 0; i < 5; ++i) d5[i] = i < d.size() ? d(i) : 0.0f;
}
}  // namespace

Points2D Calibrator::Undistort(const Points2D& img_points) {
  float K9[9], d5[5];
  flatten(K_, dist_, K9, d5);
  std::vector<float> in(img_points.size() * 2), out(img_points.size() * 2);
  for (size_t i = 0; i < img_points.size(); ++i) { in[2 * i] = img_points[i].x(); in[2 * i + 1] = img_points[i].y(); }
  const int rc = cc_undistort(device_, K9, d5, (int64_t)img_points.size(), in.data(), out.data());
  if (rc != 0) throw std::runtime_error(std::string("Calibrator::Undistort: ") + cc_last_error());
  Points2D res;
  for (size_t i = 0; i < img_points.size(); ++i) res.emplace_back(out[2 * i], out[2 * i + 1]);
  return res;
}

Points2D Calibrator::Distort(const Points2D& normalized_points) {
  float K9[9], d5[5];
  flatten(K_, dist_, K9, d5);
  std::vector<float> in(normalized_points.size() * 2), out(normalized_points.size() * 2);
  for (size_t i = 0; i < normalized_points.size(); ++i) { in[2 * i] = normalized_points[i].x(); in[2 * i + 1] = normalized_points[i].y(); }
  const int rc = cc_distort(device_, K9, d5, (int64_t)normalized_points.size(), in.data(), out.data());
  if (rc != 0) throw std::runtime_error(std::string("Calibrator::Distort: ") + cc_last_error());
  Points2D res;
  for (size_t i = 0; i < normalized_points.size(); ++i) res.emplace_back(out[2 * i], out[2 * i + 1]);
  return res;
}

}  // namespace calibrator
