// rig_scenario.cpp -- synthetic multi-camera rig scenario of the reference's own rig test
// (/root/reference/src/test_extrinsics_calibrator.cpp:9-134) at any size, behind the harness C ABI
// (include/cc_harness.h). Harness code: bench.py and the tests build their rig inputs with it; the solver never
// calls it. std::mt19937 and std::uniform_real_distribution<float> as in the reference, draws in the order the
// test makes them (the three angle draws of DistortTransformation sit in one C++ expression whose operand order
// is unspecified: taken left to right here).
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "../../include/cc_harness.h"
#include "geometry.hh"
#include "types.hh"

namespace calibrator {
namespace {

using Mat4 = Eigen::Matrix4f;

// linear part of `a` times the rotation about one coordinate axis (0 = x, 1 = y, 2 = z); float arithmetic
Eigen::Affine3f rotate_right(const Eigen::Affine3f& a, int axis, float angle) {
  const float c = std::cos(angle), s = std::sin(angle);
  float r[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
  const int i = (axis + 1) % 3, j = (axis + 2) % 3;   // the plane the rotation acts in
  r[i][i] = c; r[i][j] = -s; r[j][i] = s; r[j][j] = c;
  Eigen::Affine3f out = a;
  for (int row = 0; row < 3; ++row)
    for (int col = 0; col < 3; ++col)
      out.matrix()(row, col) = a.matrix()(row, 0) * r[0][col] + a.matrix()(row, 1) * r[1][col] + a.matrix()(row, 2) * r[2][col];
  return out;
}

// DistortTransformation (test_extrinsics_calibrator.cpp:9-38): T.linear() * (Rz(a) * Ry(b) * Rz(c)), then the
// translation is shifted by three draws
Eigen::Affine3f perturb(const Eigen::Affine3f& T, std::mt19937& gen, float translation_error, float rotation_error_deg) {
  std::uniform_real_distribution<float> shift(-translation_error, translation_error);
  std::uniform_real_distribution<float> angle_deg(-rotation_error_deg, rotation_error_deg);
  constexpr float pi = 3.141592653589793f;
  Eigen::Affine3f err = Eigen::Affine3f::Identity();
  err = rotate_right(err, 2, angle_deg(gen) / 180.0f * pi);
  err = rotate_right(err, 1, angle_deg(gen) / 180.0f * pi);
  err = rotate_right(err, 2, angle_deg(gen) / 180.0f * pi);
  Eigen::Affine3f out = T;
  for (int row = 0; row < 3; ++row)
    for (int col = 0; col < 3; ++col)
      out.matrix()(row, col) = T.matrix()(row, 0) * err.matrix()(0, col) + T.matrix()(row, 1) * err.matrix()(1, col) +
                               T.matrix()(row, 2) * err.matrix()(2, col);
  for (int i = 0; i < 3; ++i) out.matrix()(i, 3) = T.matrix()(i, 3) + shift(gen);
  return out;
}

Point3D apply(const Eigen::Affine3f& T, const Point3D& p) {
  const Mat4& m = T.matrix();
  return Point3D(m(0, 0) * p.x() + m(0, 1) * p.y() + m(0, 2) * p.z() + m(0, 3), m(1, 0) * p.x() + m(1, 1) * p.y() + m(1, 2) * p.z() + m(1, 3),
                 m(2, 0) * p.x() + m(2, 1) * p.y() + m(2, 2) * p.z() + m(2, 3));
}

Point3D unit(const Point3D& p) {
  const float n = std::sqrt(p.x() * p.x() + p.y() * p.y() + p.z() * p.z());
  return n > 0.0f ? Point3D(p.x() / n, p.y() / n, p.z() / n) : p;
}
Point3D cross(const Point3D& a, const Point3D& b) {
  return Point3D(a.y() * b.z() - a.z() * b.y(), a.z() * b.x() - a.x() * b.z(), a.x() * b.y() - a.y() * b.x());
}

}  // namespace
}  // namespace calibrator

extern "C" {

// cam_T / cam_T_true: 16 floats per camera, frame_T: 16 per frame, column-major like the JSON wire format
// (extrinsics_calibrator.cpp:271); world_xyz: 3 floats per world point (frames x pts); observations point-major
// inside a frame, one per camera (the AddObservation order of the test): obs_cam/obs_world/obs_uv have
// cams * frames * pts entries. cam_T is what the test hands to AddCameraTRig (camera 0 exact and frozen, the
// others perturbed by 5 mm / 0.1 deg), frame_T what it hands to AddObservationFrame (perturbed by 20 mm / 1 deg).
void cc_rig_scenario(int32_t cams, int32_t frames, int32_t pts, uint32_t seed, float* cam_T, float* cam_T_true,
                     float* frame_T, float* world_xyz, uint32_t* obs_cam, uint64_t* obs_world, float* obs_uv) {
  using namespace calibrator;
  std::mt19937 gen{seed};
  auto store = [](const Eigen::Affine3f& T, float* out) { for (int i = 0; i < 16; ++i) out[i] = T.matrix()(i); };
  std::uniform_real_distribution<float> rig_offset(-0.03f, 0.03f);
  std::vector<Eigen::Affine3f> truth((size_t)cams);
  for (int c = 1; c < cams; ++c) {               // test_extrinsics_calibrator.cpp:62-68
    truth[(size_t)c].matrix()(0, 3) = rig_offset(gen);
    truth[(size_t)c].matrix()(1, 3) = rig_offset(gen);
  }
  for (int c = 0; c < cams; ++c) {               // :70-83
    store(truth[(size_t)c], cam_T_true + 16 * c);
    store(c == 0 ? truth[0] : perturb(truth[(size_t)c], gen, 0.005f, 0.1f), cam_T + 16 * c);
  }
  std::uniform_real_distribution<float> rig_position(0.3f, 1.0f), point_coord(-0.2f, 0.2f);
  const float err2 = 2.0f / 500.0f, err3 = 0.001f;
  std::uniform_real_distribution<float> noise_2d(-err2, err2), noise_3d(-err3, err3);
  int64_t n_world = 0, n_obs = 0;
  for (int f = 0; f < frames; ++f) {             // :93-134
    Eigen::Affine3f rig_T_world = Eigen::Affine3f::Identity();
    Point3D t;
    t.x() = rig_position(gen); t.y() = rig_position(gen); t.z() = rig_position(gen);
    const Point3D forward = unit(t);
    const Point3D right = unit(cross(Point3D(0.0f, 1.0f, 0.0f), forward));
    const Point3D up = cross(forward, right);
    for (int i = 0; i < 3; ++i) {
      rig_T_world.matrix()(0, i) = forward[i];
      rig_T_world.matrix()(1, i) = right[i];
      rig_T_world.matrix()(2, i) = up[i];
      rig_T_world.matrix()(i, 3) = t[i];
    }
    store(perturb(rig_T_world, gen, 0.02f, 1.0f), frame_T + 16 * f);
    for (int p = 0; p < pts; ++p) {
      Point3D X;
      X.x() = point_coord(gen); X.y() = point_coord(gen); X.z() = point_coord(gen);
      Point3D Xn = X;
      Xn.x() += noise_3d(gen); Xn.y() += noise_3d(gen); Xn.z() += noise_3d(gen);
      for (int i = 0; i < 3; ++i) world_xyz[3 * n_world + i] = Xn[i];
      for (int c = 0; c < cams; ++c) {
        const Point3D in_cam = apply(truth[(size_t)c], apply(rig_T_world, X));
        float u = in_cam.x() / in_cam.z(), v = in_cam.y() / in_cam.z();
        u += noise_2d(gen);
        v += noise_2d(gen);
        obs_cam[n_obs] = (uint32_t)c;
        obs_world[n_obs] = (uint64_t)n_world;
        obs_uv[2 * n_obs] = u;
        obs_uv[2 * n_obs + 1] = v;
        ++n_obs;
      }
      ++n_world;
    }
  }
}

// Affine3f (16 floats, column-major) -> the fp64 quaternion (w x y z) and translation ExtrinsicsCalibrator::Optimize
// starts from (extrinsics_calibrator.cpp:116-130)
void cc_affine_to_qt(const float* T16, double* q_wxyz, double* t_xyz) {
  Eigen::Affine3f T;
  for (int i = 0; i < 16; ++i) T.matrix()(i) = T16[i];
  calibrator::AffineToQuaternionTranslation(T, q_wxyz, t_xyz);
}

}  // extern "C"
