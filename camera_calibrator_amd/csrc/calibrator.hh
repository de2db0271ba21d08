// calibrator.hh -- single-camera calibration with the public surface of the reference's
// calibrator::Calibrator (src/calibrator.hh:8-52). The bundle adjustment behind Optimize runs on
// the GPU through the C ABI of include/cc_solver.h instead of Ceres.
#pragma once
#include <set>
#include <vector>

#include "types.hh"

struct cc_summary;

namespace calibrator {

class Calibrator {
 public:
  Calibrator(const int img_width, const int img_height);

  /// OpenCV's cv::calibrateCamera wrapper of the reference (calibrator.cpp:16-45). OpenCV is not
  /// part of this build: throws std::runtime_error.
  void EstimateOpenCv(const std::vector<Points2D>& in_img_points,
                      const std::vector<Points3D>& in_world_points);

  /// Zhang initialisation (homographies -> K -> poses) followed by Optimize (calibrator.cpp:47-68).
  void Estimate(const std::vector<Points2D>& in_img_points,
                const std::vector<Points3D>& in_world_points);

  /// Reprojection-error bundle adjustment over the 9 intrinsics and one pose per image
  /// (calibrator.cpp:221-336). K and the distortion are updated; like the reference, the refined
  /// poses are not written back to qs / ts.
  void Optimize(const std::vector<Points2D>& in_img_points,
                const std::vector<Points3D>& in_world_points, std::vector<Quaternion>& qs,
                std::vector<Point3D>& ts);

  Matrix3 GetK() const { return K_; }
  DynamicVector GetDistortion() const { return dist_; }
  void SetK(const Matrix3& K) { K_ = K; }
  void SetDistortion(const DynamicVector& dist) { dist_ = dist; }

  /// Freeze distortion coefficient `distortion_idx` (order k1 k2 p1 p2 k3) during Optimize.
  void ForceDistortionToConstant(const int distortion_idx);

  /// Pixel coordinates -> undistorted normalised coordinates (calibrator.cpp:118-155).
  Points2D Undistort(const Points2D& img_points);
  /// Normalised coordinates -> distorted pixel coordinates (calibrator.cpp:157-166).
  Points2D Distort(const Points2D& normalized_points);

  // ---- additions of this build (not in the reference) ----
  /// GPU used by Optimize / Distort / Undistort (default 0).
  void SetDevice(int device) { device_ = device; }
  /// Status of the last Optimize: 0 or a negative cc_status; the reference has no error channel
  /// (ceres' summary is discarded), so Optimize itself never throws on solver failure.
  int LastStatus() const { return last_status_; }
  int LastIterations() const { return last_iterations_; }
  double LastFinalCost() const { return last_final_cost_; }

 private:
  int width_;
  int height_;
  int device_{0};
  int last_status_{0};
  int last_iterations_{0};
  double last_final_cost_{0.0};
  Matrix3 K_{Matrix3::Identity()};
  DynamicVector dist_{DynamicVector::Zero(5)};
  std::set<int> constant_intrinsics_;
};

}  // namespace calibrator
