// Native drop-in check of the C++ class surface: no Python, no pybind11 -- host C++ calling the HIP
// solver through the C ABI, the way a C++ user of the reference links against it.
// Scenarios after the reference's own tests (constants cited), assertions stronger than theirs:
//   1. src/test_calibrator.cpp:11-21,45-72   Calibrator::Estimate on 5 planar views, K within 1 %
//   2. src/test_extrinsics_calibrator.cpp:48-139  two-camera rig, Serialize -> Parse -> Optimize
// Exit code 0 = all checks passed; every failed check prints a line.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "calibrator.hh"
#include "data_generator.hh"
#include "extrinsics_calibrator.hh"
#include "cc_harness.h"

using namespace calibrator;

static int g_failed = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) { ++g_failed; std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

// ---- tiny helpers on Affine3f (column-major 4x4) ---------------------------------------------
static Eigen::Affine3f compose(const Eigen::Affine3f& A, const Eigen::Affine3f& B) {
  Eigen::Affine3f C;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A.matrix()(r, k) * B.matrix()(k, c);
      C.matrix()(r, c) = s;
    }
  return C;
}
static void apply(const Eigen::Affine3f& T, const float p[3], float out[3]) {
  for (int r = 0; r < 3; ++r)
    out[r] = T.matrix()(r, 0) * p[0] + T.matrix()(r, 1) * p[1] + T.matrix()(r, 2) * p[2] + T.matrix()(r, 3);
}
static Eigen::Affine3f axis_angle(float ax, float ay, float az, float angle) {
  const float n = std::sqrt(ax * ax + ay * ay + az * az);
  Eigen::Affine3f T;
  if (n == 0.f) return T;
  const float x = ax / n, y = ay / n, z = az / n, c = std::cos(angle), s = std::sin(angle), v = 1.f - c;
  const float R[9] = {c + x * x * v, x * y * v - z * s, x * z * v + y * s, y * x * v + z * s, c + y * y * v,
                      y * z * v - x * s, z * x * v - y * s, z * y * v + x * s, c + z * z * v};
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) T.matrix()(r, cc) = R[r * 3 + cc];
  return T;
}
// small random rigid perturbation: translation within +-trans per axis, rotation within +-deg
static Eigen::Affine3f perturb(const Eigen::Affine3f& T, std::mt19937& gen, float trans, float deg) {
  std::uniform_real_distribution<float> ut(-trans, trans), ua(-1.f, 1.f), ud(-deg, deg);
  Eigen::Affine3f D = axis_angle(ua(gen), ua(gen), ua(gen), ud(gen) * 3.14159265f / 180.f);
  D.matrix()(0, 3) = ut(gen); D.matrix()(1, 3) = ut(gen); D.matrix()(2, 3) = ut(gen);
  return compose(D, T);
}
static float translation_error(const Eigen::Affine3f& A, const Eigen::Affine3f& B) {
  float e = 0.f;
  for (int r = 0; r < 3; ++r) e = std::fmax(e, std::fabs(A.matrix()(r, 3) - B.matrix()(r, 3)));
  return e;
}

// ---- 1. single camera ---------------------------------------------------------------------------
static void single_camera() {
  const int w = 1600, h = 1000;                        // test_calibrator.cpp:13
  Matrix3 K = Matrix3::Zero();
  K(0, 0) = 1000.f; K(1, 1) = 1000.f; K(0, 2) = w / 2.0f; K(1, 2) = h / 2.0f; K(2, 2) = 1.f;   // :14-17
  DynamicVector dist(5);
  const float truth[5] = {-4.0e-2f, 5e-4f, 1.0e-3f, 2.0e-5f, -3e-4f};                           // :19
  for (int i = 0; i < 5; ++i) dist(i) = truth[i];
  DataGenerator generator(w, h);
  generator.SetK(K);
  generator.SetDistortion(dist);
  generator.SetNoiseInPixels(0.5f);                                                              // :21
  std::vector<Points2D> img;
  std::vector<Points3D> world;
  for (int view = 0; view < 5; ++view) {                                                         // :47-58
    GeneratedData p = generator.GetDistortedPointsPlanar(100);
    CHECK(p.image.size() == 100 && p.world.size() == 100);
    img.push_back(p.image);
    world.push_back(p.world);
  }
  Calibrator calibrator(w, h);
  calibrator.Estimate(img, world);
  const Matrix3 Kn = calibrator.GetK();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      if (K(r, c) != 0.f) CHECK(std::fabs(Kn(r, c) - K(r, c)) / K(r, c) < 0.01f);               // :62-71 (two-sided)
      else CHECK(Kn(r, c) == 0.f);
    }
  const DynamicVector d = calibrator.GetDistortion();
  CHECK(d.size() == 5);
  std::printf("single camera: fx %.3f fy %.3f px %.3f py %.3f  dist %.5f %.5f %.5f %.5f %.5f\n", Kn(0, 0), Kn(1, 1),
              Kn(0, 2), Kn(1, 2), d(0), d(1), d(2), d(3), d(4));
  CHECK(std::fabs(d(0) - truth[0]) < 2e-2f && std::fabs(d(2) - truth[2]) < 5e-3f && std::fabs(d(3) - truth[3]) < 5e-3f);
  // Distort / Undistort are inverse to each other on the image (calibrator.cpp:97-166)
  Points2D norm;
  for (int i = 0; i < 50; ++i) norm.push_back(Point2D(-0.5f + 0.02f * i, 0.3f - 0.01f * i));
  const Points2D pix = calibrator.Distort(norm);
  const Points2D back = calibrator.Undistort(pix);
  CHECK(pix.size() == norm.size() && back.size() == norm.size());
  float worst = 0.f;
  for (size_t i = 0; i < norm.size(); ++i)
    worst = std::fmax(worst, std::fmax(std::fabs(back[i](0) - norm[i](0)), std::fabs(back[i](1) - norm[i](1))));
  CHECK(worst < 1e-5f);
}

// ---- 2. two-camera rig ---------------------------------------------------------------------------
static void rig() {
  const int num_cams = 2, num_frames = 200, pts_per_frame = 4;   // the reference uses 1000 frames (:57)
  std::mt19937 gen(0);
  std::uniform_real_distribution<float> cam_xy(-0.03f, 0.03f), frame_t(0.3f, 1.0f), pt(-0.2f, 0.2f),
      err2d(-2.0f / 500.0f, 2.0f / 500.0f), err3d(-0.001f, 0.001f);                              // :60-66,90-95
  std::vector<Eigen::Affine3f> cams_true(num_cams), cams_start(num_cams);
  ExtrinsicsCalibrator calib;
  calib.SetVerbose(false);
  for (int c = 0; c < num_cams; ++c) {
    if (c > 0) { cams_true[c].matrix()(0, 3) = cam_xy(gen); cams_true[c].matrix()(1, 3) = cam_xy(gen); }
    cams_start[c] = c == 0 ? cams_true[c] : perturb(cams_true[c], gen, 0.005f, 0.1f);          // :61-62
    const size_t id = calib.AddCameraTRig(cams_start[c], c == 0);
    CHECK(id == (size_t)c);
  }
  size_t n_obs = 0;
  for (int f = 0; f < num_frames; ++f) {
    // rig looks at the origin from t (rows of the rotation: forward, right, up)   :99-112
    const float t[3] = {frame_t(gen), frame_t(gen), frame_t(gen)};
    const float tn = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    const float fw[3] = {t[0] / tn, t[1] / tn, t[2] / tn};
    float rt[3] = {fw[2], 0.f, -fw[0]};   // (0,1,0) x forward
    const float rn = std::sqrt(rt[0] * rt[0] + rt[2] * rt[2]);
    rt[0] /= rn; rt[2] /= rn;
    const float up[3] = {fw[1] * rt[2] - fw[2] * rt[1], fw[2] * rt[0] - fw[0] * rt[2], fw[0] * rt[1] - fw[1] * rt[0]};
    Eigen::Affine3f rig_T_world;
    for (int c = 0; c < 3; ++c) {
      rig_T_world.matrix()(0, c) = fw[c]; rig_T_world.matrix()(1, c) = rt[c]; rig_T_world.matrix()(2, c) = up[c];
      rig_T_world.matrix()(c, 3) = t[c];
    }
    const size_t fid = calib.AddObservationFrame(perturb(rig_T_world, gen, 0.02f, 1.0f));       // :63-64,114-117
    CHECK(fid == (size_t)f);
    for (int k = 0; k < pts_per_frame; ++k) {
      const float X[3] = {pt(gen), pt(gen), pt(gen)};
      const size_t wid = calib.AddWorldPoint(fid, Point3D(X[0] + err3d(gen), X[1] + err3d(gen), X[2] + err3d(gen)));
      CHECK(wid == (size_t)(f * pts_per_frame + k));
      for (int c = 0; c < num_cams; ++c) {
        float xr[3], xc[3];
        apply(rig_T_world, X, xr);
        apply(cams_true[c], xr, xc);
        calib.AddObservation(c, wid, Point2D(xc[0] / xc[2] + err2d(gen), xc[1] / xc[2] + err2d(gen)));
        ++n_obs;
      }
    }
  }
  const std::string fname = "/tmp/cc_dropin_rig.json";
  calib.Serialize(fname);                                                                        // :136-139
  // Parse into the SAME object appends the cameras again and rebuilds the frozen set from the parsed
  // ones only (the reference does not clear camera_T_rigs_, extrinsics_calibrator.cpp:349-351): kept.
  ExtrinsicsCalibrator quirk = calib;
  quirk.Parse(fname);
  CHECK(quirk.NumCameras() == 4 && !quirk.IsCameraFrozen(0) && quirk.IsCameraFrozen(2));
  // ... so a reader starts from a fresh object
  ExtrinsicsCalibrator loaded;
  loaded.SetVerbose(false);
  loaded.Parse(fname);
  CHECK(loaded.NumCameras() == 2 && loaded.IsCameraFrozen(0) && !loaded.IsCameraFrozen(1));
  CHECK(loaded.NumObservationFrames() == (size_t)num_frames && loaded.NumWorldPoints() == (size_t)(num_frames * pts_per_frame));
  calib = loaded;
  calib.Optimize();
  CHECK(calib.LastStatus() == 0);
  CHECK(calib.LastSolverReruns() == 0 && calib.LastSolverNote().empty());
  CHECK(translation_error(calib.GetCameraTRig(0), cams_true[0]) == 0.f);   // frozen camera untouched
  const float before = translation_error(cams_start[1], cams_true[1]);
  const float after = translation_error(calib.GetCameraTRig(1), cams_true[1]);
  std::printf("rig: %zu observations, %d iterations, final cost %.6g, camera 1 translation error %.2e -> %.2e\n", n_obs,
              calib.LastIterations(), calib.LastFinalCost(), before, after);
  CHECK(after < 2.5e-3f && after < before);
  size_t cam = 0, widx = 0, wid = 0;
  Point2D uv;
  double cost = 0.0;
  calib.GetObservation(3, 5, &cam, &widx, &wid, &uv, &cost);
  CHECK(std::isfinite(cost) && cost >= 0.0 && cam < 2 && wid == 3 * pts_per_frame + widx);
  // bookkeeping after removing frames (extrinsics_calibrator.cpp:415-452)
  calib.RemoveObservationFrames({0, 7, 3});
  CHECK(calib.NumObservationFrames() == (size_t)(num_frames - 3));
  CHECK(calib.NumWorldPoints() == (size_t)((num_frames - 3) * pts_per_frame));
  // ... and the solve of what is left is the solve of the same data read into a fresh object (the class keeps the world points
  // flat by global id, which a removal invalidates; the JSON numbers are shortest round-trip)
  const std::string fname2 = "/tmp/cc_dropin_rig_removed.json";
  calib.Serialize(fname2);
  ExtrinsicsCalibrator fresh;
  fresh.SetVerbose(false);
  fresh.Parse(fname2);
  calib.Optimize();
  fresh.Optimize();
  CHECK(calib.LastStatus() == 0 && fresh.LastStatus() == 0);
  CHECK(calib.LastIterations() == fresh.LastIterations() && calib.LastFinalCost() == fresh.LastFinalCost());
  CHECK(translation_error(calib.GetCameraTRig(1), fresh.GetCameraTRig(1)) == 0.f);
}

// ---- 3. what a drop-in user pays per call: Calibrator::Estimate at BASELINE configs[2] size through the class ----------------
// (bench.py embeds this line as `class_surface`; --class-surface F M REPS)
#include <chrono>
#include <cstring>
static int class_surface(int frames, int pts, int reps) {
  const int w = 1600, h = 1000;
  Matrix3 K = Matrix3::Zero();
  K(0, 0) = 1000.f; K(1, 1) = 1000.f; K(0, 2) = w / 2.0f; K(1, 2) = h / 2.0f; K(2, 2) = 1.f;
  DynamicVector dist(5);
  const float truth[5] = {-4.0e-2f, 5e-4f, 1.0e-3f, 2.0e-5f, -3e-4f};
  for (int i = 0; i < 5; ++i) dist(i) = truth[i];
  DataGenerator generator(w, h);
  generator.SetK(K);
  generator.SetDistortion(dist);
  generator.SetNoiseInPixels(0.5f);
  std::vector<Points2D> img;
  std::vector<Points3D> world;
  for (int view = 0; view < frames; ++view) {
    GeneratedData p = generator.GetDistortedPointsPlanar(pts);
    img.push_back(p.image);
    world.push_back(p.world);
  }
  std::vector<double> total, parts[7];
  int iters = 0;
  double first_ms = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    // a fresh object per call, as the reference's workflow makes one (cam_calibration.py:306-309): what is reused between
    // calls is the library's cached device arena, pinned staging block, stream and control block
    Calibrator c(w, h);
    const auto t0 = std::chrono::steady_clock::now();
    c.Estimate(img, world);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    iters = c.LastIterations();
    if (r == 0) { first_ms = ms; continue; }   // (first call: allocations, code object load)
    total.push_back(ms);
    for (int k = 0; k < 7; ++k) parts[k].push_back(c.LastTimingMs()[k]);
  }
  auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  std::printf("{\"call\": \"Calibrator::Estimate through the C++ class (fresh object per call)\", \"frames\": %d, \"pts\": %d, \"observations\": %d, "
              "\"lm_iterations\": %d, \"calls\": %d, \"first_call_ms\": %.4f, \"wall_ms_median\": %.4f, \"class_side_pack_ms\": %.4f, \"handle_and_arena_ms\": %.4f, "
              "\"pack_and_upload_ms\": %.4f, \"zhang_ms\": %.4f, \"solve_ms\": %.4f, \"readback_ms\": %.4f}\n",
              frames, pts, frames * pts, iters, reps, first_ms, median(total), median(parts[0]), median(parts[1]), median(parts[2]),
              median(parts[3]), median(parts[4]), median(parts[5]));
  return 0;
}

// ---- 4. the same for the rig: ExtrinsicsCalibrator::Optimize, every camera sees every point (--class-surface-rig C F M REPS) ----
static int class_surface_rig(int num_cams, int num_frames, int pts_per_frame, int reps) {
  // the product harness's scenario (include/cc_harness.h: the reference's rig test at any size) -- what bench.py's `configs` solve
  const size_t C = (size_t)num_cams, F = (size_t)num_frames, M = (size_t)pts_per_frame, N = C * F * M;
  std::vector<float> cam_T(16 * C), cam_T_true(16 * C), frame_T(16 * F), world(3 * F * M), uv(2 * N);
  std::vector<uint32_t> obs_cam(N);
  std::vector<uint64_t> obs_world(N);
  cc_rig_scenario(num_cams, num_frames, pts_per_frame, 0u, cam_T.data(), cam_T_true.data(), frame_T.data(), world.data(), obs_cam.data(),
                  obs_world.data(), uv.data());
  auto affine = [](const float* T16) { Eigen::Affine3f T; for (int i = 0; i < 16; ++i) T.matrix()(i % 4, i / 4) = T16[i]; return T; };
  ExtrinsicsCalibrator base;
  base.SetVerbose(false);
  for (size_t c = 0; c < C; ++c) base.AddCameraTRig(affine(&cam_T[16 * c]), c == 0);
  for (size_t f = 0, k = 0; f < F; ++f) {
    const size_t fid = base.AddObservationFrame(affine(&frame_T[16 * f]));
    for (size_t j = 0; j < M; ++j) base.AddWorldPoint(fid, Point3D(world[3 * (f * M + j)], world[3 * (f * M + j) + 1], world[3 * (f * M + j) + 2]));
    for (size_t j = 0; j < M * C; ++j, ++k) base.AddObservation(obs_cam[k], (size_t)obs_world[k], Point2D(uv[2 * k], uv[2 * k + 1]));
  }
  auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  std::vector<double> parts[4], again[4];
  int iters = 0, iters_again = 0;
  double first_ms = 0.0;
  for (int r = 0; r < reps + 1; ++r) {
    ExtrinsicsCalibrator c = base;   // (a copy of the starting state; the copy is outside the clock)
    c.Optimize();
    if (r == 0) { first_ms = c.LastTimingMs()[3]; continue; }   // (first call: allocations, code object load)
    iters = c.LastIterations();
    for (int k = 0; k < 4; ++k) parts[k].push_back(c.LastTimingMs()[k]);
    c.Optimize();                    // the same object again: its flat arrays are warm, the solve starts at its optimum
    iters_again = c.LastIterations();
    for (int k = 0; k < 4; ++k) again[k].push_back(c.LastTimingMs()[k]);
  }
  std::printf("{\"call\": \"ExtrinsicsCalibrator::Optimize through the C++ class\", \"cams\": %d, \"frames\": %d, \"pts\": %d, \"observations\": %lld, "
              "\"calls\": %d, \"first_call_ms\": %.3f, "
              "\"fresh_object\": {\"lm_iterations\": %d, \"wall_ms_median\": %.3f, \"flatten_ms\": %.3f, \"cc_rig_optimize_ms\": %.3f, \"write_back_ms\": %.3f}, "
              "\"same_object_again\": {\"lm_iterations\": %d, \"wall_ms_median\": %.3f, \"flatten_ms\": %.3f, \"cc_rig_optimize_ms\": %.3f, \"write_back_ms\": %.3f}}\n",
              num_cams, num_frames, pts_per_frame, (long long)num_cams * num_frames * pts_per_frame, reps, first_ms,
              iters, median(parts[3]), median(parts[0]), median(parts[1]), median(parts[2]),
              iters_again, median(again[3]), median(again[0]), median(again[1]), median(again[2]));
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::strcmp(argv[1], "--class-surface-rig") == 0)
    return class_surface_rig(argc > 2 ? std::atoi(argv[2]) : 4, argc > 3 ? std::atoi(argv[3]) : 400, argc > 4 ? std::atoi(argv[4]) : 300,
                             argc > 5 ? std::atoi(argv[5]) : 5);
  if (argc >= 2 && std::strcmp(argv[1], "--class-surface") == 0)
    return class_surface(argc > 2 ? std::atoi(argv[2]) : 1000, argc > 3 ? std::atoi(argv[3]) : 500, argc > 4 ? std::atoi(argv[4]) : 20);
  single_camera();
  rig();
  std::printf(g_failed ? "%d check(s) FAILED\n" : "all checks passed\n", g_failed);
  return g_failed ? 1 : 0;
}
