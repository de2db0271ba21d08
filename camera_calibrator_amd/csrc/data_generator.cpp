// data_generator.cpp -- see data_generator.hh. Draw order and float/double promotions follow
// /root/reference/src/data_generator.cpp line by line in behaviour (cited below), not in code.
#include "data_generator.hh"

#include <cmath>

#include "../../include/cc_harness.h"
#include "geometry.hh"

namespace calibrator {

namespace {
inline Point3D mul(const Matrix3& M, const Point3D& p) {
  return Point3D(M(0, 0) * p.x() + M(0, 1) * p.y() + M(0, 2) * p.z(), M(1, 0) * p.x() + M(1, 1) * p.y() + M(1, 2) * p.z(),
                 M(2, 0) * p.x() + M(2, 1) * p.y() + M(2, 2) * p.z());
}
inline Point3D scaled_unit(const Point3D& p, float s) {  // p.normalize(); p *= s  (Eigen float semantics)
  const float z = p.x() * p.x() + p.y() * p.y() + p.z() * p.z();
  Point3D o = p;
  if (z > 0.0f) { const float n = std::sqrt(z); o = Point3D(p.x() / n, p.y() / n, p.z() / n); }
  return Point3D(o.x() * s, o.y() * s, o.z() * s);
}
}  // namespace

namespace {
using Camera = DataGenerator::Camera;
using Draws = DataGenerator::Draws;

// pixel of a camera-frame point through the distortion model, plus noise; false when the noisy pixel
// leaves the image (data_generator.cpp:10-32; the two noise draws happen before the test)
bool project_noisy(const Camera& cam, Draws& rnd, const Point3D& p, Point2D* out) {
  const double x = (double)p.x() / (double)p.z(), y = (double)p.y() / (double)p.z();
  const double k1 = cam.dist(0), k2 = cam.dist(1), p1 = cam.dist(2), p2 = cam.dist(3), k3 = cam.dist(4);
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
  const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  const double xd = x * cdist + p1 * a1 + p2 * a2, yd = y * cdist + p1 * a3 + p2 * a1;
  const float u = (float)(xd * (double)cam.K(0, 0) + (double)cam.K(0, 2));
  const float v = (float)(yd * (double)cam.K(1, 1) + (double)cam.K(1, 2));
  const float ud = u + rnd.pixel_noise(rnd.engine);
  const float vd = v + rnd.pixel_noise(rnd.engine);
  if (ud < 0.0f || ud >= cam.width - 1 || vd < 0.0f || vd >= cam.height - 1) return false;
  *out = Point2D(ud, vd);
  return true;
}

// a point on the viewing ray of a uniformly drawn pixel, at a uniformly drawn distance
// (data_generator.cpp:126-146: u, v, then the distance)
Point3D frustum_point(Draws& rnd, const Matrix3& K_inv) {
  const float u = rnd.u(rnd.engine);
  const float v = rnd.v(rnd.engine);
  const Point3D ray = mul(K_inv, Point3D(u, v, 1.0f));
  const float depth = rnd.depth(rnd.engine);
  return scaled_unit(ray, depth);
}

// plane through three image corners pushed out to random distances (data_generator.cpp:52-75)
Plane corner_plane(const Camera& cam, Draws& rnd, const Matrix3& K_inv) {
  Point3D corner(0.0f, 0.0f, 1.0f);
  const Point3D a = scaled_unit(mul(K_inv, corner), rnd.depth(rnd.engine));
  corner.x() = static_cast<float>(cam.width - 1);
  const Point3D b = scaled_unit(mul(K_inv, corner), rnd.depth(rnd.engine));
  corner.y() = static_cast<float>(cam.height - 1);
  const Point3D c = scaled_unit(mul(K_inv, corner), rnd.depth(rnd.engine));
  return EstimatePlaneFinite(a, b, c);
}
}  // namespace

DataGenerator::DataGenerator(int img_width, int img_height) {
  cam_.width = img_width;
  cam_.height = img_height;
  rnd_.u = std::uniform_real_distribution<float>(0.0f, img_width - 1.0f);
  rnd_.v = std::uniform_real_distribution<float>(0.0f, img_height - 1.0f);
}

void DataGenerator::SetK(const Matrix3& K) { cam_.K = K; }
void DataGenerator::SetDistortion(const DynamicVector& dist) { cam_.dist = dist; }
void DataGenerator::SetNoiseInPixels(const float noise) {
  rnd_.pixel_noise = std::uniform_real_distribution<float>(-noise, noise);
}

// data_generator.cpp:77-124
GeneratedData DataGenerator::GetDistortedPointsPlanar(const int num_p) {
  GeneratedData out;
  const Matrix3 K_inv = Inverse3x3(cam_.K);  // pseudo-inverse of an invertible K
  const Plane plane = corner_plane(cam_, rnd_, K_inv);
  const Matrix3 R = RotationMatrixFromPlane(plane);
  while ((int)out.image.size() < num_p) {
    const Point3D p = frustum_point(rnd_, K_inv);
    const Point3D on_plane = ProjectToPlane(plane, p, p);
    const Point3D rotated = mul(R, on_plane);
    Point2D px;
    if (!project_noisy(cam_, rnd_, on_plane, &px)) { ++rejected_; continue; }
    out.world.emplace_back(rotated.x(), rotated.y(), 0.0f);
    out.image.push_back(px);
  }
  return out;
}

// data_generator.cpp:148-184
GeneratedData DataGenerator::GetDistortedPoints(const int num_p) {
  GeneratedData out;
  const Matrix3 K_inv = Inverse3x3(cam_.K);
  while ((int)out.image.size() < num_p) {
    const Point3D p = frustum_point(rnd_, K_inv);
    Point2D px;
    if (!project_noisy(cam_, rnd_, p, &px)) { ++rejected_; continue; }
    out.world.push_back(p);
    out.image.push_back(px);
  }
  return out;
}

}  // namespace calibrator

// ---- C ABI (include/cc_harness.h) ---------------------------------------------------------------
struct cc_generator {
  calibrator::DataGenerator gen;
  cc_generator(int w, int h) : gen(w, h) {}
};

extern "C" {

cc_generator* cc_generator_create(int32_t width, int32_t height) { return new cc_generator(width, height); }
void cc_generator_destroy(cc_generator* g) { delete g; }
void cc_generator_set_k(cc_generator* g, const float* K9) {
  calibrator::Matrix3 K;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K(r, c) = K9[r * 3 + c];
  g->gen.SetK(K);
}
void cc_generator_set_distortion(cc_generator* g, const float* dist5) {
  calibrator::DynamicVector d = calibrator::DynamicVector::Zero(5);
  for (int i = 0; i < 5; ++i) d(i) = dist5[i];
  g->gen.SetDistortion(d);
}
void cc_generator_set_noise(cc_generator* g, float noise) { g->gen.SetNoiseInPixels(noise); }

static int64_t emit(const calibrator::GeneratedData& d, float* uv, float* xyz) {
  for (size_t i = 0; i < d.image.size(); ++i) {
    uv[2 * i] = d.image[i].x(); uv[2 * i + 1] = d.image[i].y();
    xyz[3 * i] = d.world[i].x(); xyz[3 * i + 1] = d.world[i].y(); xyz[3 * i + 2] = d.world[i].z();
  }
  return (int64_t)d.image.size();
}
int64_t cc_generator_planar(cc_generator* g, int32_t num_p, float* uv, float* xyz) {
  return emit(g->gen.GetDistortedPointsPlanar(num_p), uv, xyz);
}
int64_t cc_generator_points(cc_generator* g, int32_t num_p, float* uv, float* xyz) {
  return emit(g->gen.GetDistortedPoints(num_p), uv, xyz);
}

}  // extern "C"
