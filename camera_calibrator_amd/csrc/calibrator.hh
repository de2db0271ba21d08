// calibrator.hh -- single-camera calibration with the public surface of the reference's
// calibrator::Calibrator (src/calibrator.hh:8-52). The bundle adjustment behind Optimize runs on
// the GPU through the C ABI of include/cc_solver.h instead of Ceres.
#pragma once
#include <set>
#include <string>
#include <vector>

#include "types.hh"

struct cc_summary;

namespace calibrator {

class Calibrator {
 public:
  Calibrator(const int image_width, const int image_height);

  // ---- the model: K (fx, fy, px, py; zero skew) and the distortion k1 k2 p1 p2 k3 ---------------
  Matrix3 GetK() const { return camera_matrix_; }
  void SetK(const Matrix3& K) { camera_matrix_ = K; }
  DynamicVector GetDistortion() const { return distortion_; }
  void SetDistortion(const DynamicVector& dist) { distortion_ = dist; }
  /// Keep distortion coefficient `coefficient` (0..4 = k1 k2 p1 p2 k3) at its current value in Optimize.
  void ForceDistortionToConstant(const int coefficient);

  // ---- estimation ---------------------------------------------------------------------------------
  /// Closed-form start (homography per view -> K -> board poses) followed by Optimize
  /// (reference: calibrator.cpp:47-68). The closed form runs on the GPU too (cc_zhang_init).
  void Estimate(const std::vector<Points2D>& pixels_per_view, const std::vector<Points3D>& board_points_per_view);
  /// The reference's cv::calibrateCamera wrapper (calibrator.cpp:16-45). OpenCV is not part of this build; the call computes what
  /// calibrateCamera(flags = 0) minimises -- the same objective over the same nine parameters as Estimate(), nothing held
  /// constant, distortion started from zero -- with this library's Zhang initialisation + bundle adjustment (round 6; it threw
  /// before). Same minimiser, not OpenCV's trajectory; parity unpinned.
  void EstimateOpenCv(const std::vector<Points2D>& pixels_per_view, const std::vector<Points3D>& board_points_per_view);
  /// Reprojection-error bundle adjustment over the 9 intrinsics and one pose per view, starting from
  /// the given poses (reference: calibrator.cpp:221-336). Updates K and the distortion; like the
  /// reference it leaves qs / ts as they were.
  void Optimize(const std::vector<Points2D>& pixels_per_view, const std::vector<Points3D>& board_points_per_view,
                std::vector<Quaternion>& qs, std::vector<Point3D>& ts);

  // ---- mapping points through the model ------------------------------------------------------------
  /// pixels -> undistorted normalised coordinates (reference: calibrator.cpp:118-155)
  Points2D Undistort(const Points2D& pixels);
  /// normalised coordinates -> distorted pixels (reference: calibrator.cpp:157-166)
  Points2D Distort(const Points2D& normalised);

  // ---- additions of this build (not in the reference) ----------------------------------------------
  /// GPU used by Optimize / Estimate / Distort / Undistort (default 0).
  void SetDevice(int device) { device_ = device; devices_.clear(); }
  /// Several GPUs for Optimize (and the Optimize inside Estimate): the views are sharded over them and ONE host
  /// thread drives all of them (cc_intrinsics_optimize_multi). The first one also serves the point kernels.
  void SetDevices(const std::vector<int>& devices) { devices_ = devices; if (!devices.empty()) device_ = devices[0]; }
  /// Status of the last Optimize: 0 or a negative cc_status. The reference has no error channel
  /// (ceres' summary is discarded), so Optimize itself never throws on solver failure.
  int LastStatus() const { return last_status_; }
  int LastIterations() const { return last_iterations_; }
  /// Did the last call's solve have to be run AGAIN in another form of the solver (cc_last_call_solver_status: the persistent
  /// one-launch kernel gave up because its workgroups were not resident together -- another tenant on the GPU, a tool that
  /// serialises kernels, another host thread inside a device-wide runtime call)? > 0: that many times; the call was late by 42 ms
  /// to 1.3 s each and its result equals the usual one to rounding only. LastSolverNote() says what the kernel reported.
  int LastSolverReruns() const { return last_solver_reruns_; }
  /// The form the last call's solve ran in to its end: 0 several kernels per LM iteration, 1 / 2 / 4 the persistent per-solve
  /// kernel(s). 0 on a device that holds persistent solves means the device's back-off window after a give-up (the next 8 solves
  /// and 2 s, doubling on every further give-up; one solve then probes the persistent form again).
  int LastSolverForm() const { return last_solver_form_; }
  const std::string& LastSolverNote() const { return last_solver_note_; }
  double LastFinalCost() const { return last_final_cost_; }
  /// Wall milliseconds of the last Estimate / Optimize: [0] packing the views into (cached, pinned) flat arrays when the
  /// class does it (several devices; on one device the library packs piece by piece under its upload and the time is
  /// part of [2]), [1] solver handle + device arena (cached), [2] pack + upload, [3] Zhang initialisation (Estimate),
  /// [4] solve, [5] read-back + teardown, [6] the whole call.
  const double* LastTimingMs() const { return last_timing_ms_; }

 private:
  void ReadSolverStatus();
  int image_w_;
  int image_h_;
  int device_{0};
  std::vector<int> devices_;
  int last_status_{0};
  int last_iterations_{0};
  int last_solver_reruns_{0};
  int last_solver_form_{0};
  std::string last_solver_note_;
  double last_final_cost_{0.0};
  double last_timing_ms_[7]{0, 0, 0, 0, 0, 0, 0};
  Matrix3 camera_matrix_{Matrix3::Identity()};
  DynamicVector distortion_{DynamicVector::Zero(5)};
  std::set<int> frozen_intrinsics_;
};

}  // namespace calibrator
