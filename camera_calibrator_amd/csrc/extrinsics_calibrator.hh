// extrinsics_calibrator.hh -- multi-camera rig pose calibration with the public surface of the
// reference's calibrator::ExtrinsicsCalibrator (src/extrinsics_calibrator.hh:8-70). Optimize runs on
// the GPU through cc_rig_optimize (include/cc_solver.h) instead of Ceres.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "types.hh"

namespace calibrator {

class ExtrinsicsCalibrator {
 public:
  /// Registers a camera pose (camera_T_rig). Frozen cameras keep their pose. Returns the camera id.
  size_t AddCameraTRig(const Eigen::Affine3f& camera_T_rig, const bool freeze = false);
  Eigen::Affine3f GetCameraTRig(const size_t id) const;

  /// Registers the initial rig pose (pose) of a new observation frame. Returns the frame id.
  size_t AddObservationFrame(const Eigen::Affine3f& pose);
  Eigen::Affine3f GetObservationFrame(const size_t id) const;

  /// Adds a world point to a frame. Returns the (global) world point id.
  size_t AddWorldPoint(const size_t frame_id, const Point3D& world_point);

  /// Adds an observation of a world point, in normalised (undistorted) image coordinates.
  void AddObservation(const size_t camera, const size_t point_global, const Point2D& normalised);

  /// Huber-robustified reprojection-error bundle adjustment over all non-frozen camera poses and
  /// all frame poses (extrinsics_calibrator.cpp:86-257). Updates the stored transforms and the
  /// per-observation costs; prints one progress line per iteration like the reference does.
  void Optimize();

  void Serialize(const std::string& fname) const;
  void Parse(const std::string& fname);

  void RemoveObservationFrame(const size_t frame);
  void RemoveObservationFrames(const std::vector<size_t> observation_frame_ids);

  // ---- additions of this build (not in the reference) ----
  void SetDevice(int device) { device_ = device; devices_.clear(); }
  /// Several GPUs: the observation frames are sharded over them, one host thread drives all (cc_rig_optimize_multi).
  void SetDevices(const std::vector<int>& devices) { devices_ = devices; if (!devices.empty()) device_ = devices[0]; }
  void SetVerbose(bool verbose) { verbose_ = verbose; }
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
  /// Wall milliseconds of the last Optimize(): [0] preparing the arguments (several devices: flattening the frames), [1] cc_rig_optimize_frames
  /// (regrouping, upload, solve, per-observation costs, read-back), [2] writing costs and poses back, [3] the whole call.
  const double* LastTimingMs() const { return last_timing_ms_; }
  /// Bookkeeping introspection used by the tests (ids are what the reference's private members hold).
  size_t NumCameras() const { return cameras_.size(); }
  size_t NumObservationFrames() const { return frames_.size(); }
  size_t NumWorldPoints() const { return point_refs_.size(); }
  bool IsCameraFrozen(size_t id) const { return frozen_.count(id) != 0; }
  size_t NumObservations(size_t frame_id) const { return frames_[frame_id].NumObservations(); }
  /// (camera, point_in_frame, point_global, half_rho) of observation k of a frame
  void GetObservation(size_t frame_id, size_t k, size_t* camera, size_t* point_in_frame,
                      size_t* point_global, Point2D* normalised, double* half_rho) const;

 private:
  struct Frame {
    explicit Frame(const Eigen::Affine3f& T) : pose(T) {}
    Eigen::Affine3f pose;
    Points3D points;
    // The observations of the frame's points as COLUMNS, one entry each per observation (round 5; a record per observation
    // before): the solve streams camera + point + image point (20 bytes) per observation and writes the costs as one array --
    // at BASELINE configs[4] size (8 M observations) 160 MB read and 64 MB written where 40-byte records cost 320 MB read and
    // 320 MB read-modified-written (cc_rig_optimize_columns reads and writes these arrays in place).
    std::vector<uint32_t> obs_camera;          // camera id as the solve reads it: 4 bytes, ids beyond 2^32 - 2 (never a valid camera) as 2^32 - 1 ...
    std::map<size_t, size_t> obs_camera_wide;  // ... and, for exactly those, observation index -> the id AddObservation was given (GetObservation and
                                               // Serialize return what came in, like the reference's size_t member, extrinsics_calibrator.hh:58)
    std::vector<uint32_t> obs_point_in_frame;  // index inside the frame's points (AddWorldPoint refuses a 2^32-th point in one frame: 48 GB of them)
    std::vector<uint64_t> obs_point_global;    // global id
    std::vector<Point2D, Eigen::aligned_allocator<Point2D>> obs_normalised;
    std::vector<double> obs_half_rho;
    size_t NumObservations() const { return obs_camera.size(); }
    size_t CameraOf(size_t k) const {
      if (obs_camera[k] != 0xFFFFFFFFu) return obs_camera[k];
      const auto it = obs_camera_wide.find(k);
      return it == obs_camera_wide.end() ? (size_t)obs_camera[k] : it->second;
    }
  };
  struct PointRef {
    size_t frame;
    size_t point_in_frame;
  };

  std::vector<Eigen::Affine3f, Eigen::aligned_allocator<Eigen::Affine3f>> cameras_;
  std::vector<Frame, Eigen::aligned_allocator<Frame>> frames_;
  std::vector<PointRef> point_refs_;
  std::vector<float> world_flat_;   // x y z of every world point by global id (what cc_rig_optimize_* takes), appended by AddWorldPoint
  bool world_flat_stale_{false};    // a frame was removed: rebuilt from the frames by the next Optimize
  std::set<size_t> frozen_;
  int device_{0};
  std::vector<int> devices_;
  bool verbose_{true};
  int last_status_{0};
  int last_iterations_{0};
  int last_solver_reruns_{0};
  int last_solver_form_{0};
  std::string last_solver_note_;
  double last_final_cost_{0.0};
  // flat copies of the observations for the C ABI, kept between calls (grow-only: a caller that optimises again after adding
  // frames -- the reference's workflow -- does not fault in 200 MB of fresh vectors per call at BASELINE configs[4] size)
  // (not copied with the object; handed to the next object when this one goes: extrinsics_calibrator.cpp)
  struct FlatArrays {
    std::vector<uint32_t> cam;
    std::vector<uint64_t> world;
    std::vector<float> uv;
    std::vector<double> rho;
    FlatArrays() = default;
    FlatArrays(const FlatArrays&) {}
    FlatArrays& operator=(const FlatArrays&) { return *this; }
    ~FlatArrays();
  };
  FlatArrays flat_;
  double last_timing_ms_[4]{0, 0, 0, 0};
};

}  // namespace calibrator
