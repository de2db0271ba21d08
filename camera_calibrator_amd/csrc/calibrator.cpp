// calibrator.cpp -- Calibrator on top of the C ABI (include/cc_solver.h).
#include "calibrator.hh"

#include <cassert>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <stdexcept>

#include "../../include/cc_solver.h"
#include "geometry.hh"

namespace calibrator {

namespace {
// parameter order of the shared block, identical to the reference's enum (calibrator.cpp:168-179)
enum Intrinsic { FX, FY, PX, PY, K1, K2, P1, P2, K3, kNumIntrinsics };

// The reference copies every point into Ceres parameter blocks (calibrator.cpp:261-292); here the views are packed into the
// flat arrays of the C ABI -- ONE memcpy per view (a Points2D / Points3D is a contiguous array of two / three floats per
// point, with Eigen as with mini_eigen.hh) straight into pinned host memory that is cached between calls
// (cc_host_staging_acquire): the upload then runs at the full PCIe rate and no per-point loop, push_back or pageable
// bounce buffer is left on the path of a caller that re-estimates as images arrive (cam_calibration.py:290-322).
static_assert(sizeof(Point2D) == 2 * sizeof(float) && sizeof(Point3D) == 3 * sizeof(float), "points are packed float arrays");
struct PackedViews {
  void* block = nullptr;
  int64_t* offsets = nullptr;
  float* uv = nullptr;
  float* xyz = nullptr;
  int64_t n_obs = 0;
  double pack_ms = 0.0;
  PackedViews(const std::vector<Points2D>& pix, const std::vector<Points3D>& pts) {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n_img = pix.size();
    size_t n = 0;
    for (size_t i = 0; i < n_img; ++i) n += pix[i].size();
    const size_t b_off = ((n_img + 1) * sizeof(int64_t) + 255) & ~(size_t)255, b_uv = (n * 2 * sizeof(float) + 255) & ~(size_t)255;
    block = cc_host_staging_acquire(b_off + b_uv + n * 3 * sizeof(float) + 256);
    if (!block) throw std::runtime_error("Calibrator: pinned staging memory could not be allocated");
    offsets = static_cast<int64_t*>(block);
    uv = reinterpret_cast<float*>(static_cast<char*>(block) + b_off);
    xyz = reinterpret_cast<float*>(static_cast<char*>(block) + b_off + b_uv);
    offsets[0] = 0;
    for (size_t i = 0; i < n_img; ++i) {
      assert(pix[i].size() == pts[i].size());
      const size_t k = (size_t)offsets[i], m = pix[i].size();
      if (m) {
        std::memcpy(uv + 2 * k, pix[i].data(), m * sizeof(Point2D));
        std::memcpy(xyz + 3 * k, pts[i].data(), m * sizeof(Point3D));
      }
      offsets[i + 1] = (int64_t)(k + m);
    }
    n_obs = (int64_t)n;
    pack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  void release() { cc_host_staging_release(block); block = nullptr; }   // (before another packing of the same thread: the cache holds one block)
  ~PackedViews() { if (block) cc_host_staging_release(block); }
  PackedViews(const PackedViews&) = delete;
  PackedViews& operator=(const PackedViews&) = delete;
};
}  // namespace

Calibrator::Calibrator(const int image_width, const int image_height) : image_w_(image_width), image_h_(image_height) {}

// What LastSolverReruns() / LastSolverNote() report: the status of the library call this thread has just made
// (cc_last_call_solver_status) -- a persistent solve that gave up and was run again in the several-kernel form shows here.
void Calibrator::ReadSolverStatus() {
  char note[640] = "";
  int32_t form = 0, reruns = 0;
  cc_last_call_solver_status(&form, &reruns, note, (int32_t)sizeof(note));
  last_solver_reruns_ = reruns;
  last_solver_form_ = form;
  last_solver_note_ = note;
}

// The reference's EstimateOpenCv (calibrator.cpp:16-45) hands the views to cv::calibrateCamera with flags = 0 and keeps K and the
// five distortion coefficients it returns. OpenCV is not part of this build. What that call computes is the minimiser of the SAME
// objective over the SAME nine parameters as Estimate() -- pinhole + (k1, k2, p1, p2, k3), every coefficient free, the distortion
// started from zero, K and the poses initialised from the views' homographies -- so it is served by this library's own path:
// Zhang initialisation + bundle adjustment with NOTHING held constant (calibrateCamera knows nothing of ForceDistortionToConstant)
// and the current distortion ignored (flags = 0 carries no CALIB_USE_INTRINSIC_GUESS). Same minimiser, another trajectory and
// stopping rule than OpenCV's LM; parity unpinned (there is no OpenCV here to compare with). Throws only where Estimate() does.
void Calibrator::EstimateOpenCv(const std::vector<Points2D>& pixels_per_view, const std::vector<Points3D>& board_points_per_view) {
  (void)image_w_;   // (cv::calibrateCamera takes the image size for its initial principal point; Zhang's closed form needs none)
  (void)image_h_;
  const std::set<int> held = frozen_intrinsics_;
  frozen_intrinsics_.clear();
  distortion_ = DynamicVector::Zero(5);
  try {
    Estimate(pixels_per_view, board_points_per_view);
  } catch (...) {
    frozen_intrinsics_ = held;
    throw;
  }
  frozen_intrinsics_ = held;
}

void Calibrator::Estimate(const std::vector<Points2D>& pixels_per_view, const std::vector<Points3D>& board_points_per_view) {
  assert(pixels_per_view.size() == board_points_per_view.size());
  const size_t n_img = pixels_per_view.size();
  // Zhang initialisation on the device (cc_zhang_init): homographies -> K -> poses
  const auto t_call = std::chrono::steady_clock::now();
  for (double& v : last_timing_ms_) v = 0.0;
  float K9[9];
  if (devices_.size() <= 1) {
    // one device: the fused entry point (one upload of the observations for initialisation and solve together);
    // same numbers as the two calls below exchange through camera_matrix_ and the float poses. The views go over as they
    // are (one pointer per view): the library packs them into pinned memory piece by piece under the upload.
    double intr[kNumIntrinsics], dist5[5];
    for (int i = 0; i < 5; ++i) dist5[i] = distortion_(i);
    uint32_t frozen = 0;
    for (int idx : frozen_intrinsics_)
      if (idx >= 0 && idx < kNumIntrinsics) frozen |= 1u << idx;
    std::vector<double> qd(4 * n_img), td(3 * n_img);
    std::vector<const float*> uv_views(n_img), xyz_views(n_img);
    std::vector<int64_t> counts(n_img);
    for (size_t i = 0; i < n_img; ++i) {
      assert(pixels_per_view[i].size() == board_points_per_view[i].size());
      uv_views[i] = reinterpret_cast<const float*>(pixels_per_view[i].data());
      xyz_views[i] = reinterpret_cast<const float*>(board_points_per_view[i].data());
      counts[i] = (int64_t)pixels_per_view[i].size();
    }
    cc_options options;
    cc_options_init(&options);  // non-monotonic steps, 100 iterations: calibrator.cpp:314-321
    cc_summary summary{};
    last_status_ = cc_intrinsics_estimate_views(&options, device_, (int64_t)n_img, uv_views.data(), xyz_views.data(), counts.data(),
                                                dist5, frozen, K9, intr, qd.data(), td.data(), &summary);
    cc_last_call_timing(&last_timing_ms_[1]);
    // Same contract as the two-step path below: environment errors (no device, HIP, exchange) and the Zhang
    // preconditions (cc_zhang_init's CC_ERR_BAD_ARGUMENT: < 3 frames, < 4 points in a frame) throw; a solver-level
    // status (CC_ERR_STATE: the LM loop gave up) does not -- it goes to LastStatus() as Optimize() documents, and the
    // class holds Zhang's K plus whatever point the solver reached, as after cc_zhang_init + Optimize. The
    // reference's Estimate never throws (calibrator.cpp:47-68).
    if (last_status_ == CC_ERR_NO_DEVICE || last_status_ == CC_ERR_HIP || last_status_ == CC_ERR_COMM || last_status_ == CC_ERR_BAD_ARGUMENT)
      throw std::runtime_error(std::string("Calibrator::Estimate: ") + cc_last_error());
    last_iterations_ = summary.iterations;
    ReadSolverStatus();
    last_final_cost_ = summary.final_cost;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) camera_matrix_(r, c) = K9[r * 3 + c];
    camera_matrix_(0, 0) = static_cast<float>(intr[FX]);
    camera_matrix_(1, 1) = static_cast<float>(intr[FY]);
    camera_matrix_(0, 2) = static_cast<float>(intr[PX]);
    camera_matrix_(1, 2) = static_cast<float>(intr[PY]);
    distortion_(0) = static_cast<float>(intr[K1]);
    distortion_(1) = static_cast<float>(intr[K2]);
    distortion_(2) = static_cast<float>(intr[P1]);
    distortion_(3) = static_cast<float>(intr[P2]);
    distortion_(4) = static_cast<float>(intr[K3]);
    last_timing_ms_[6] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    return;
  }
  PackedViews pv(pixels_per_view, board_points_per_view);
  const int64_t* offsets_p = pv.offsets;
  const float *uv_p = pv.uv, *xyz_p = pv.xyz;
  last_timing_ms_[0] = pv.pack_ms;
  std::vector<float> q(4 * n_img), t(3 * n_img);
  const int rc = cc_zhang_init(device_, (int64_t)n_img, offsets_p, uv_p, xyz_p, K9, q.data(), t.data(), nullptr);
  if (rc != 0) throw std::runtime_error(std::string("Calibrator::Estimate: ") + cc_last_error());
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) camera_matrix_(r, c) = K9[r * 3 + c];
  std::vector<Quaternion> qs(n_img);
  std::vector<Point3D> ts(n_img);
  for (size_t i = 0; i < n_img; ++i) {
    qs[i] = Quaternion(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);
    ts[i] = Point3D(t[3 * i], t[3 * i + 1], t[3 * i + 2]);
  }
  pv.release();
  Optimize(pixels_per_view, board_points_per_view, qs, ts);
}

void Calibrator::Optimize(const std::vector<Points2D>& pixels_per_view, const std::vector<Points3D>& board_points_per_view,
                          std::vector<Quaternion>& qs, std::vector<Point3D>& ts) {
  const size_t n_img = pixels_per_view.size();
  assert(n_img == board_points_per_view.size() && n_img == qs.size() && n_img == ts.size());
  // fp64 parameter arrays; the ragged views go over as they are on one device (the library packs them under its upload),
  // as one CSR layout in cached pinned memory (one memcpy per view) for several devices
  const auto t_call = std::chrono::steady_clock::now();
  for (double& v : last_timing_ms_) v = 0.0;
  std::vector<double> q(4 * n_img), t(3 * n_img);
  for (size_t i = 0; i < n_img; ++i) {
    q[4 * i] = qs[i].w(); q[4 * i + 1] = qs[i].x(); q[4 * i + 2] = qs[i].y(); q[4 * i + 3] = qs[i].z();
    t[3 * i] = ts[i].x(); t[3 * i + 1] = ts[i].y(); t[3 * i + 2] = ts[i].z();
  }
  double intr[kNumIntrinsics];
  intr[FX] = camera_matrix_(0, 0); intr[FY] = camera_matrix_(1, 1); intr[PX] = camera_matrix_(0, 2); intr[PY] = camera_matrix_(1, 2);
  intr[K1] = distortion_(0); intr[K2] = distortion_(1); intr[P1] = distortion_(2); intr[P2] = distortion_(3); intr[K3] = distortion_(4);
  uint32_t frozen = 0;
  for (int idx : frozen_intrinsics_)
    if (idx >= 0 && idx < kNumIntrinsics) frozen |= 1u << idx;

  cc_options options;
  cc_options_init(&options);  // non-monotonic steps, 100 iterations: calibrator.cpp:314-321
  cc_summary summary{};
  if (n_img == 0) {
    last_status_ = 0;
  } else if (devices_.size() > 1) {
    PackedViews pv(pixels_per_view, board_points_per_view);
    last_timing_ms_[0] = pv.pack_ms;
    std::vector<int32_t> devs(devices_.begin(), devices_.end());
    last_status_ = cc_intrinsics_optimize_multi(&options, (int32_t)devs.size(), devs.data(), (int64_t)n_img, pv.offsets, pv.uv,
                                                pv.xyz, intr, frozen, q.data(), t.data(), &summary);
  } else {
    std::vector<const float*> uv_views(n_img), xyz_views(n_img);
    std::vector<int64_t> counts(n_img);
    for (size_t i = 0; i < n_img; ++i) {
      assert(pixels_per_view[i].size() == board_points_per_view[i].size());
      uv_views[i] = reinterpret_cast<const float*>(pixels_per_view[i].data());
      xyz_views[i] = reinterpret_cast<const float*>(board_points_per_view[i].data());
      counts[i] = (int64_t)pixels_per_view[i].size();
    }
    last_status_ = cc_intrinsics_optimize_views(&options, device_, (int64_t)n_img, uv_views.data(), xyz_views.data(), counts.data(),
                                                intr, frozen, q.data(), t.data(), &summary);
    cc_last_call_timing(&last_timing_ms_[1]);
  }
  last_iterations_ = summary.iterations;
  last_final_cost_ = summary.final_cost;
  if (n_img == 0) {
    last_solver_reruns_ = 0;
    last_solver_form_ = 0;
    last_solver_note_.clear();
  } else {
    ReadSolverStatus();  // (the several-device route included: cc_intrinsics_optimize_multi records its call the same way)
  }
  if (last_status_ == CC_ERR_NO_DEVICE || last_status_ == CC_ERR_HIP || last_status_ == CC_ERR_COMM)
    throw std::runtime_error(std::string("Calibrator::Optimize: ") + cc_last_error());  // no silent CPU path

  camera_matrix_(0, 0) = static_cast<float>(intr[FX]);
  camera_matrix_(1, 1) = static_cast<float>(intr[FY]);
  camera_matrix_(0, 2) = static_cast<float>(intr[PX]);
  camera_matrix_(1, 2) = static_cast<float>(intr[PY]);
  distortion_(0) = static_cast<float>(intr[K1]);
  distortion_(1) = static_cast<float>(intr[K2]);
  distortion_(2) = static_cast<float>(intr[P1]);
  distortion_(3) = static_cast<float>(intr[P2]);
  distortion_(4) = static_cast<float>(intr[K3]);
  last_timing_ms_[6] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
}

void Calibrator::ForceDistortionToConstant(const int coefficient) {
  frozen_intrinsics_.insert(coefficient + static_cast<int>(K1));
}

namespace {
void flatten(const Matrix3& K, const DynamicVector& d, float* K9, float* d5) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K9[r * 3 + c] = K(r, c);
  for (int i = 0; i < 5; ++i) d5[i] = i < d.size() ? d(i) : 0.0f;
}
}  // namespace

Points2D Calibrator::Undistort(const Points2D& pixels) {
  float K9[9], d5[5];
  flatten(camera_matrix_, distortion_, K9, d5);
  std::vector<float> in(pixels.size() * 2), out(pixels.size() * 2);
  for (size_t i = 0; i < pixels.size(); ++i) { in[2 * i] = pixels[i].x(); in[2 * i + 1] = pixels[i].y(); }
  const int rc = cc_undistort(device_, K9, d5, (int64_t)pixels.size(), in.data(), out.data());
  if (rc != 0) throw std::runtime_error(std::string("Calibrator::Undistort: ") + cc_last_error());
  Points2D res;
  for (size_t i = 0; i < pixels.size(); ++i) res.emplace_back(out[2 * i], out[2 * i + 1]);
  return res;
}

Points2D Calibrator::Distort(const Points2D& normalised) {
  float K9[9], d5[5];
  flatten(camera_matrix_, distortion_, K9, d5);
  std::vector<float> in(normalised.size() * 2), out(normalised.size() * 2);
  for (size_t i = 0; i < normalised.size(); ++i) { in[2 * i] = normalised[i].x(); in[2 * i + 1] = normalised[i].y(); }
  const int rc = cc_distort(device_, K9, d5, (int64_t)normalised.size(), in.data(), out.data());
  if (rc != 0) throw std::runtime_error(std::string("Calibrator::Distort: ") + cc_last_error());
  Points2D res;
  for (size_t i = 0; i < normalised.size(); ++i) res.emplace_back(out[2 * i], out[2 * i + 1]);
  return res;
}

}  // namespace calibrator
