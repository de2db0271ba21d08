// extrinsics_calibrator.cpp -- ExtrinsicsCalibrator on top of the C ABI (include/cc_solver.h).
#include "extrinsics_calibrator.hh"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "../../include/cc_solver.h"
#include "geometry.hh"
#include "json_min.hh"

namespace calibrator {

// ---- id bookkeeping (reference: extrinsics_calibrator.cpp:9-49) -------------------------------

size_t ExtrinsicsCalibrator::AddCameraTRig(const Eigen::Affine3f& camera_T_rig, const bool freeze) {
  const size_t id = cameras_.size();
  cameras_.push_back(camera_T_rig);
  if (freeze) frozen_.insert(id);
  return id;
}

Eigen::Affine3f ExtrinsicsCalibrator::GetCameraTRig(const size_t id) const { return cameras_[id]; }

size_t ExtrinsicsCalibrator::AddObservationFrame(const Eigen::Affine3f& pose) {
  frames_.emplace_back(pose);
  return frames_.size() - 1;
}

Eigen::Affine3f ExtrinsicsCalibrator::GetObservationFrame(const size_t id) const { return frames_[id].pose; }

size_t ExtrinsicsCalibrator::AddWorldPoint(const size_t frame_id, const Point3D& world_point) {
  Frame& frame = frames_[frame_id];
  if (frame.points.size() >= 0xFFFFFFFFull) throw std::length_error("ExtrinsicsCalibrator::AddWorldPoint: a frame holds at most 2^32 - 1 world points");
  point_refs_.push_back(PointRef{frame_id, frame.points.size()});
  frame.points.push_back(world_point);
  world_flat_.push_back(world_point.x()); world_flat_.push_back(world_point.y()); world_flat_.push_back(world_point.z());
  return point_refs_.size() - 1;
}

void ExtrinsicsCalibrator::AddObservation(const size_t camera, const size_t point_global, const Point2D& normalised) {
  const PointRef& info = point_refs_[point_global];
  Frame& frame = frames_[info.frame];  // stored with the point's frame
  if (camera >= 0xFFFFFFFFull) frame.obs_camera_wide[frame.obs_camera.size()] = camera;   // (kept as given for GetObservation / Serialize)
  frame.obs_camera.push_back(camera < 0xFFFFFFFFull ? (uint32_t)camera : 0xFFFFFFFFu);
  frame.obs_point_in_frame.push_back((uint32_t)info.point_in_frame);
  frame.obs_point_global.push_back((uint64_t)point_global);
  frame.obs_normalised.push_back(normalised);
  frame.obs_half_rho.push_back(std::numeric_limits<double>::quiet_NaN());  // "not evaluated yet"
}

void ExtrinsicsCalibrator::GetObservation(size_t frame_id, size_t k, size_t* camera, size_t* point_in_frame,
                                          size_t* point_global, Point2D* normalised, double* half_rho) const {
  const Frame& fr = frames_[frame_id];
  if (camera) *camera = fr.CameraOf(k);
  if (point_in_frame) *point_in_frame = fr.obs_point_in_frame[k];
  if (point_global) *point_global = (size_t)fr.obs_point_global[k];
  if (normalised) *normalised = fr.obs_normalised[k];
  if (half_rho) *half_rho = fr.obs_half_rho[k];
}

// ---- the solve -----------------------------------------------------------------------------------

namespace {
// flat observation arrays of the last ExtrinsicsCalibrator that was destroyed (one set per process, at most 2^27 observations):
// the next object's first Optimize() starts with memory that is already faulted in
// (never destroyed: an ExtrinsicsCalibrator with static storage in another translation unit may go after this one's statics)
std::mutex& g_flat_mu = *new std::mutex;
std::vector<uint32_t>& g_flat_cam = *new std::vector<uint32_t>;
std::vector<uint64_t>& g_flat_world = *new std::vector<uint64_t>;
std::vector<float>& g_flat_uv = *new std::vector<float>;
std::vector<double>& g_flat_rho = *new std::vector<double>;
}  // namespace

ExtrinsicsCalibrator::FlatArrays::~FlatArrays() {
  std::lock_guard<std::mutex> lk(g_flat_mu);
  if (cam.size() > g_flat_cam.size() && cam.size() <= ((size_t)1 << 27)) {
    cam.swap(g_flat_cam); world.swap(g_flat_world); uv.swap(g_flat_uv); rho.swap(g_flat_rho);
  }
}

void ExtrinsicsCalibrator::Optimize() {
  const auto t_call = std::chrono::steady_clock::now();
  const size_t C = cameras_.size(), F = frames_.size(), Pn = point_refs_.size();
  // fp64 parameter arrays exactly as the reference fills them (extrinsics_calibrator.cpp:116-137)
  std::vector<double> cam_q(4 * C), cam_t(3 * C), frame_q(4 * F), frame_t(3 * F);
  for (size_t i = 0; i < C; ++i) AffineToQuaternionTranslation(cameras_[i], &cam_q[4 * i], &cam_t[3 * i]);
  for (size_t i = 0; i < F; ++i) AffineToQuaternionTranslation(frames_[i].pose, &frame_q[4 * i], &frame_t[3 * i]);
  // the world points by global id, as the solve takes them: kept up to date by AddWorldPoint (no gather, no fresh 12 MB vector per
  // call at BASELINE configs[4] size); a removed frame leaves it to be rebuilt here
  // (parallel phases run on the library's process-lifetime worker pool, cc_parallel_for: no thread is created per call)
  auto run_parts = [](size_t parts, const std::function<void(size_t)>& fn) {
    struct Ctx { const std::function<void(size_t)>* fn; } ctx{&fn};
    cc_parallel_for((int32_t)parts, [](void* c, int32_t t) { (*static_cast<Ctx*>(c)->fn)((size_t)t); }, &ctx);
  };
  if (world_flat_stale_ || world_flat_.size() != 3 * Pn) {
    world_flat_.resize(3 * Pn);
    float* wf = world_flat_.data();
    const size_t parts = (size_t)cc_parallel_parts((int64_t)Pn, (int64_t)1 << 16);
    run_parts(parts, [&](size_t t) {
      for (size_t i = Pn * t / parts, b = Pn * (t + 1) / parts; i < b; ++i) {
        const PointRef& info = point_refs_[i];
        const Point3D& p = frames_[info.frame].points[info.point_in_frame];
        wf[3 * i] = p.x(); wf[3 * i + 1] = p.y(); wf[3 * i + 2] = p.z();
      }
    });
    world_flat_stale_ = false;
  }
  const std::vector<float>& world = world_flat_;
  std::vector<int64_t> offsets(F + 1, 0);
  for (size_t f = 0; f < F; ++f) offsets[f + 1] = offsets[f] + (int64_t)frames_[f].NumObservations();
  const size_t N = (size_t)offsets[F];
  const bool several = devices_.size() > 1;
  // frames in contiguous ranges of about equal observation counts, one host thread each (flattening for several devices)
  std::vector<size_t> part_first{0};
  {
    const size_t parts = (size_t)cc_parallel_parts((int64_t)N, (int64_t)1 << 17);
    for (size_t t = 1; t < parts; ++t)
      part_first.push_back((size_t)(std::lower_bound(offsets.begin(), offsets.end(), (int64_t)(N * t / parts)) - offsets.begin()));
    part_first.push_back(F);
    for (size_t t = 1; t < part_first.size(); ++t) part_first[t] = std::min(F, std::max(part_first[t], part_first[t - 1]));
  }
  auto over_frames = [&](const std::function<void(size_t, size_t)>& fn) {
    run_parts(part_first.size() - 1, [&](size_t t) { fn(part_first[t], part_first[t + 1]); });
  };
  uint32_t* obs_cam = nullptr;
  uint64_t* obs_world = nullptr;
  float* obs_uv = nullptr;
  double* half_rho = nullptr;
  if (several) {
    // several devices: flat copies of the observations for cc_rig_optimize_multi, kept between calls
    if (flat_.cam.size() < N) {
      // a fresh object (the reference's workflow may build one per call): take over the arrays the last object left behind
      std::lock_guard<std::mutex> lk(g_flat_mu);
      if (g_flat_cam.size() > flat_.cam.size()) { flat_.cam.swap(g_flat_cam); flat_.world.swap(g_flat_world); flat_.uv.swap(g_flat_uv); flat_.rho.swap(g_flat_rho); }
    }
    if (flat_.cam.size() < N) { flat_.cam.resize(N); flat_.world.resize(N); flat_.uv.resize(2 * N); flat_.rho.resize(N); }
    obs_cam = flat_.cam.data();
    obs_world = flat_.world.data();
    obs_uv = flat_.uv.data();
    half_rho = flat_.rho.data();   // (cc_rig_optimize_multi writes every entry)
    over_frames([&](size_t f0, size_t f1) {
      for (size_t f = f0; f < f1; ++f) {
        const Frame& fr = frames_[f];
        const size_t k0 = (size_t)offsets[f], n = fr.NumObservations();
        if (n == 0) continue;
        std::memcpy(obs_cam + k0, fr.obs_camera.data(), n * sizeof(uint32_t));
        std::memcpy(obs_world + k0, fr.obs_point_global.data(), n * sizeof(uint64_t));
        std::memcpy(obs_uv + 2 * k0, fr.obs_normalised.data(), n * 2 * sizeof(float));
      }
    });
  }
  std::vector<uint8_t> frozen(C, 0);
  for (size_t id : frozen_) if (id < C) frozen[id] = 1;
  last_timing_ms_[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
  const auto t_lib = std::chrono::steady_clock::now();

  cc_options options;
  cc_options_init(&options);
  options.max_iterations = 1000;  // extrinsics_calibrator.cpp:211
  std::vector<cc_iteration> log(1001);
  cc_summary summary{};
  summary.log = log.data();
  summary.log_capacity = (int32_t)log.size();
  const double huber_a = 3.0f / 500.0f;  // extrinsics_calibrator.cpp:176 (float literal, as in the reference)
  last_status_ = 0;
  if (N > 0 && C > 0 && F > 0) {
    if (devices_.size() > 1) {
      std::vector<int32_t> devs(devices_.begin(), devices_.end());
      last_status_ = cc_rig_optimize_multi(&options, (int32_t)devs.size(), devs.data(), (int64_t)C, (int64_t)F, (int64_t)Pn, offsets.data(),
                                           obs_cam, obs_world, obs_uv, world.data(), cam_q.data(), cam_t.data(),
                                           frozen.data(), frame_q.data(), frame_t.data(), huber_a, half_rho, &summary);
    } else {
      // one device: the library reads the frames' columns where they are and writes the costs into them
      // (cc_rig_optimize_columns) -- no flat copies, no write-back loop on this side
      static_assert(sizeof(Point2D) == 2 * sizeof(float), "an image point is two floats, as cc_obs_columns reads it");
      std::vector<const void*> p_cam(F), p_world(F), p_uv(F);
      std::vector<void*> p_cost(F);
      std::vector<int64_t> counts(F);
      for (size_t f = 0; f < F; ++f) {
        Frame& fr = frames_[f];
        counts[f] = (int64_t)fr.NumObservations();
        p_cam[f] = fr.obs_camera.data(); p_world[f] = fr.obs_point_global.data(); p_uv[f] = fr.obs_normalised.data();
        p_cost[f] = fr.obs_half_rho.data();
      }
      cc_obs_columns cols{};
      cols.camera = p_cam.data(); cols.camera_stride = 4; cols.camera_width = 4;
      cols.world = p_world.data(); cols.world_stride = 8; cols.world_width = 8;
      cols.uv = p_uv.data(); cols.uv_stride = (int64_t)sizeof(Point2D);
      cols.cost = p_cost.data(); cols.cost_stride = 8;
      last_status_ = cc_rig_optimize_columns(&options, device_, (int64_t)C, (int64_t)F, (int64_t)Pn, &cols, counts.data(),
                                             world.data(), cam_q.data(), cam_t.data(), frozen.data(), frame_q.data(), frame_t.data(),
                                             huber_a, &summary);
    }
    if (last_status_ == CC_ERR_NO_DEVICE || last_status_ == CC_ERR_HIP || last_status_ == CC_ERR_BAD_ARGUMENT || last_status_ == CC_ERR_COMM)
      throw std::runtime_error(std::string("ExtrinsicsCalibrator::Optimize: ") + cc_last_error());  // no silent CPU path
  }
  last_iterations_ = summary.iterations;
  last_final_cost_ = summary.final_cost;
  {
    char note[640] = "";
    int32_t form = 0, reruns = 0;
    cc_last_call_solver_status(&form, &reruns, note, (int32_t)sizeof(note));
    last_solver_reruns_ = reruns;
    last_solver_form_ = form;
    last_solver_note_ = note;
  }
  if (verbose_ && last_solver_reruns_ > 0) std::printf("note: %s\n", last_solver_note_.c_str());
  if (verbose_) {  // the reference lets Ceres print its per-iteration table (extrinsics_calibrator.cpp:212-213)
    std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n");
    std::printf("%4d %.6e    0.00e+00\n", 0, summary.initial_cost);
    for (int i = 0; i < summary.log_len; ++i)
      std::printf("%4d %.6e %11.2e %10.2e %10.2e %9.2e %9.2e\n", i + 1, log[i].cost, log[i].cost_change, log[i].gradient_max_norm,
                  log[i].step_norm, log[i].relative_decrease, log[i].radius);
  }
  last_timing_ms_[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lib).count();
  const auto t_back = std::chrono::steady_clock::now();
  // per-observation robustified half_rho (extrinsics_calibrator.cpp:219-225)
  // A call that failed without throwing (CC_ERR_STATE, ...) leaves no costs of ANOTHER problem behind: flat_.rho may be storage
  // recycled from an earlier object, and on one device the library skips its write-back -- zeros in both cases.
  if (last_status_ != 0 && N > 0)
    over_frames([&](size_t f0, size_t f1) {
      for (size_t f = f0; f < f1; ++f) std::fill(frames_[f].obs_half_rho.begin(), frames_[f].obs_half_rho.end(), 0.0);
    });
  else if (several && N > 0 && C > 0 && F > 0)
    over_frames([&](size_t f0, size_t f1) {
      for (size_t f = f0; f < f1; ++f) {
        if (frames_[f].NumObservations()) std::memcpy(frames_[f].obs_half_rho.data(), half_rho + offsets[f], frames_[f].NumObservations() * sizeof(double));
      }
    });
  // poses back through float (extrinsics_calibrator.cpp:228-256)
  for (size_t i = 0; i < C; ++i) cameras_[i] = QuaternionTranslationToAffine(&cam_q[4 * i], &cam_t[3 * i]);
  for (size_t i = 0; i < F; ++i) frames_[i].pose = QuaternionTranslationToAffine(&frame_q[4 * i], &frame_t[3 * i]);
  const auto t_end = std::chrono::steady_clock::now();
  last_timing_ms_[2] = std::chrono::duration<double, std::milli>(t_end - t_back).count();
  last_timing_ms_[3] = std::chrono::duration<double, std::milli>(t_end - t_call).count();
}

// ---- JSON wire format (reference: extrinsics_calibrator.cpp:268-413) -----------------------------

namespace {
jsonmin::Value transform_to_json(const Eigen::Affine3f& T) {
  jsonmin::Value a = jsonmin::Value::array();
  for (int i = 0; i < 16; ++i) a.arr.push_back(jsonmin::Value::number(T.matrix()(i)));  // column-major (.reshaped())
  return a;
}
Eigen::Affine3f transform_from_json(const jsonmin::Value& a) {
  Eigen::Matrix4f M;
  for (size_t i = 0; i < a.size() && i < 16; ++i) M(static_cast<int>(i)) = (float)a.at(i).as_number();
  return Eigen::Affine3f(M);
}
}  // namespace

void ExtrinsicsCalibrator::Serialize(const std::string& fname) const {
  using jsonmin::Value;
  Value root = Value::object();
  Value cams = Value::array();
  for (size_t i = 0; i < cameras_.size(); ++i) {
    Value c = Value::object();
    c.obj["frozen"] = Value::boolean(frozen_.count(i) != 0);
    c.obj["camera_T_rig"] = transform_to_json(cameras_[i]);
    cams.arr.push_back(c);
  }
  Value wps = Value::array();
  for (const PointRef& info : point_refs_) {
    Value w = Value::object();
    w.obj["frame_id"] = Value::integer(info.frame);
    const Point3D& p = frames_[info.frame].points[info.point_in_frame];
    Value v = Value::array();
    for (int i = 0; i < 3; ++i) v.arr.push_back(Value::number(p(i)));
    w.obj["world_point"] = v;
    wps.arr.push_back(w);
  }
  Value frames = Value::array();
  for (const Frame& frame : frames_) {
    Value f = Value::object();
    f.obj["rig_T_world"] = transform_to_json(frame.pose);
    Value obs = Value::array();
    for (size_t k = 0; k < frame.NumObservations(); ++k) {
      Value e = Value::object();
      e.obj["camera_id"] = Value::integer(frame.CameraOf(k));
      e.obj["world_point_id"] = Value::integer(frame.obs_point_global[k]);
      Value ip = Value::array();
      ip.arr.push_back(Value::number(frame.obs_normalised[k].x()));
      ip.arr.push_back(Value::number(frame.obs_normalised[k].y()));
      e.obj["image_point"] = ip;
      e.obj["cost"] = Value::number(frame.obs_half_rho[k]);  // NaN -> null
      obs.arr.push_back(e);
    }
    f.obj["observations"] = obs;
    frames.arr.push_back(f);
  }
  root.obj["camera_T_rigs"] = cams;
  root.obj["world_points"] = wps;
  root.obj["observation_frames"] = frames;
  std::string text;
  jsonmin::dump(root, text);
  std::ofstream out(fname.c_str());
  out << text;
}

void ExtrinsicsCalibrator::Parse(const std::string& fname) {
  // Like the reference (extrinsics_calibrator.cpp:348-351) the camera list is NOT cleared: parsing
  // into a calibrator that already holds cameras appends the parsed ones behind them.
  frozen_.clear();
  point_refs_.clear();
  frames_.clear();
  world_flat_.clear();          // (the flat copy of the world points belongs to the frames just dropped: AddWorldPoint below refills it)
  world_flat_stale_ = false;
  std::ifstream in(fname.c_str());
  if (!in) throw std::runtime_error("ExtrinsicsCalibrator::Parse: cannot open " + fname);
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string text = ss.str();
  const jsonmin::Value root = jsonmin::Parser(text).parse();

  const jsonmin::Value& cams = root.at("camera_T_rigs");
  for (size_t i = 0; i < cams.size(); ++i)
    AddCameraTRig(transform_from_json(cams.at(i).at("camera_T_rig")), cams.at(i).at("frozen").as_bool());
  const jsonmin::Value& frames = root.at("observation_frames");
  for (size_t i = 0; i < frames.size(); ++i) AddObservationFrame(transform_from_json(frames.at(i).at("rig_T_world")));
  const jsonmin::Value& wps = root.at("world_points");
  for (size_t i = 0; i < wps.size(); ++i) {
    const jsonmin::Value& v = wps.at(i).at("world_point");
    AddWorldPoint((size_t)wps.at(i).at("frame_id").as_number(),
                  Point3D((float)v.at(0).as_number(), (float)v.at(1).as_number(), (float)v.at(2).as_number()));
  }
  for (size_t i = 0; i < frames.size(); ++i) {
    const jsonmin::Value& obs = frames.at(i).at("observations");
    for (size_t k = 0; k < obs.size(); ++k) {
      const jsonmin::Value& ip = obs.at(k).at("image_point");
      AddObservation((size_t)obs.at(k).at("camera_id").as_number(), (size_t)obs.at(k).at("world_point_id").as_number(),
                     Point2D((float)ip.at(0).as_number(), (float)ip.at(1).as_number()));  // half_rho is ignored (-> NaN)
    }
  }
}

// ---- frame removal with id renumbering (reference: extrinsics_calibrator.cpp:415-452) -------------

void ExtrinsicsCalibrator::RemoveObservationFrame(const size_t frame) {
  const size_t removed_points = frames_[frame].points.size();
  frames_.erase(frames_.begin() + (std::ptrdiff_t)frame);
  // world point ids of all later frames slide down by the removed frame's point count
  for (size_t f = frame; f < frames_.size(); ++f)
    for (uint64_t& g : frames_[f].obs_point_global) g -= removed_points;
  world_flat_stale_ = true;   // (rebuilt by the next Optimize)
  point_refs_.erase(std::remove_if(point_refs_.begin(), point_refs_.end(),
                                          [=](const PointRef& w) { return w.frame == frame; }),
                           point_refs_.end());
  for (PointRef& w : point_refs_)
    if (w.frame >= frame) --w.frame;
}

void ExtrinsicsCalibrator::RemoveObservationFrames(const std::vector<size_t> observation_frame_ids) {
  std::vector<size_t> ids = observation_frame_ids;
  std::sort(ids.begin(), ids.end(), std::greater<size_t>());  // highest first: earlier ids stay valid
  for (size_t id : ids) RemoveObservationFrame(id);
}

}  // namespace calibrator
