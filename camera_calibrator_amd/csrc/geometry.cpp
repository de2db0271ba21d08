// geometry.cpp -- host-side geometry of the calibrator surface (see geometry.hh).
#include "geometry.hh"

#include <cassert>
#include <cmath>
#include <iostream>

namespace calibrator {
namespace {

struct V3 { double x, y, z; };
inline V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 unit(const V3& a) { const double n = std::sqrt(dot(a, a)); return n > 0 ? V3{a.x / n, a.y / n, a.z / n} : a; }

// Thin one-sided Jacobi SVD of a column-major m x n matrix (n <= m is not required).
// After run(): columns of `a` hold U*Sigma, `v` (n x n, column-major) the right singular vectors.
class JacobiSvd {
 public:
  JacobiSvd(int m, int n) : m_(m), n_(n), a_((size_t)m * n, 0.0), v_((size_t)n * n, 0.0) {
    for (int i = 0; i < n; ++i) v_[(size_t)i * n + i] = 1.0;
  }
  double& at(int r, int c) { return a_[(size_t)c * m_ + r]; }
  void run() {
    for (int sweep = 0; sweep < 64; ++sweep) {
      bool changed = false;
      for (int p = 0; p + 1 < n_; ++p)
        for (int q = p + 1; q < n_; ++q) changed |= rotate(p, q);
      if (!changed) break;
    }
  }
  double sigma(int c) const {
    double s = 0;
    for (int r = 0; r < m_; ++r) s += a_[(size_t)c * m_ + r] * a_[(size_t)c * m_ + r];
    return std::sqrt(s);
  }
  int smallest() const {
    int best = 0;
    for (int c = 1; c < n_; ++c) if (sigma(c) < sigma(best)) best = c;
    return best;
  }
  double v(int r, int c) const { return v_[(size_t)c * n_ + r]; }
  double us(int r, int c) const { return a_[(size_t)c * m_ + r]; }

 private:
  bool rotate(int p, int q) {
    double* ap = &a_[(size_t)p * m_];
    double* aq = &a_[(size_t)q * m_];
    double alpha = 0, beta = 0, gamma = 0;
    for (int r = 0; r < m_; ++r) { alpha += ap[r] * ap[r]; beta += aq[r] * aq[r]; gamma += ap[r] * aq[r]; }
    if (gamma == 0.0 || std::fabs(gamma) <= 2.3e-16 * std::sqrt(alpha * beta)) return false;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
    for (int r = 0; r < m_; ++r) { const double x = ap[r], y = aq[r]; ap[r] = c * x - s * y; aq[r] = s * x + c * y; }
    double* vp = &v_[(size_t)p * n_];
    double* vq = &v_[(size_t)q * n_];
    for (int r = 0; r < n_; ++r) { const double x = vp[r], y = vq[r]; vp[r] = c * x - s * y; vq[r] = s * x + c * y; }
    return true;
  }
  int m_, n_;
  std::vector<double> a_, v_;
};

// U V^T of a 3x3 (row-major in/out, double)
void polar3(const double* M, double* out) {
  JacobiSvd svd(3, 3);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) svd.at(r, c) = M[r * 3 + c];
  svd.run();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += (svd.us(r, k) / svd.sigma(k)) * svd.v(c, k);
      out[r * 3 + c] = a;
    }
}

template <class P1, class P2>
Matrix3 homography_dlt(const P1& p1, const P2& p2) {
  // HZ2 alg. 4.1: two rows per correspondence, null vector of the 2n x 9 system (fp64)
  assert(p1.size() == p2.size());
  const int n = (int)p1.size();
  JacobiSvd svd(2 * n, 9);
  for (int i = 0; i < n; ++i) {
    const float x1 = p1[i].x(), y1 = p1[i].y(), x2 = p2[i].x(), y2 = p2[i].y();
    const int r = 2 * i;
    svd.at(r, 3) = -x1; svd.at(r, 4) = -y1; svd.at(r, 5) = -1.0;
    svd.at(r, 6) = x1 * y2; svd.at(r, 7) = y1 * y2; svd.at(r, 8) = y2;
    svd.at(r + 1, 0) = x1; svd.at(r + 1, 1) = y1; svd.at(r + 1, 2) = 1.0;
    svd.at(r + 1, 6) = -x1 * x2; svd.at(r + 1, 7) = -y1 * x2; svd.at(r + 1, 8) = -x2;
  }
  svd.run();
  const int c = svd.smallest();
  Matrix3 H;
  for (int i = 0; i < 9; ++i) H(i / 3, i % 3) = (float)svd.v(i, c);
  return H;
}

}  // namespace

Plane EstimatePlaneFinite(const Point3D& p1, const Point3D& p2, const Point3D& p3) {
  Matrix3 A;
  const Point3D* rows[3] = {&p1, &p2, &p3};
  for (int r = 0; r < 3; ++r) { A(r, 0) = rows[r]->x(); A(r, 1) = rows[r]->y(); A(r, 2) = rows[r]->z(); }
  const Matrix3 Ai = Inverse3x3(A);
  Plane out;
  for (int r = 0; r < 3; ++r) out(r) = Ai(r, 0) + Ai(r, 1) + Ai(r, 2);  // A^-1 * (1,1,1)
  out(3) = -1.0f;
  return out;
}

Point3D PlaneNormal(const Plane& plane) {
  const float n = std::sqrt(plane(0) * plane(0) + plane(1) * plane(1) + plane(2) * plane(2));
  return Point3D(plane(0) / n, plane(1) / n, plane(2) / n);
}

Matrix3 RotationMatrixFromPlane(const Plane& plane, const Point3D& new_normal) {
  if (!(std::fabs(new_normal.x()) < 1e-5f && std::fabs(new_normal.y()) < 1e-5f && std::fabs(new_normal.z() - 1.0f) < 1e-5f))
    std::cerr << "Warning: RotationMatrixFromPlane with new_normal other than UnitZ not tested" << std::endl;
  // float arithmetic throughout, as Eigen's Vector3f cross()/normalized() in geometry.cpp:29-36
  auto unitf = [](float x, float y, float z, float* o) {
    const float zz = x * x + y * y + z * z;
    if (zz > 0.0f) { const float n = std::sqrt(zz); o[0] = x / n; o[1] = y / n; o[2] = z / n; } else { o[0] = x; o[1] = y; o[2] = z; }
  };
  auto crossf = [](const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  };
  const Point3D nf = PlaneNormal(plane);
  const float n[3] = {nf.x(), nf.y(), nf.z()};
  float nn[3], c1[3], v1[3], c2[3], v2[3];
  unitf(new_normal.x(), new_normal.y(), new_normal.z(), nn);
  crossf(n, nn, c1);
  unitf(c1[0], c1[1], c1[2], v1);
  crossf(n, v1, c2);
  unitf(c2[0], c2[1], c2[2], v2);
  Matrix3 R;
  for (int c = 0; c < 3; ++c) { R(0, c) = v1[c]; R(1, c) = v2[c]; R(2, c) = n[c]; }
  return R;
}

Point3D ProjectToPlane(const Plane& plane, const Point3D& p, const std::optional<Point3D>& projection_direction) {
  // (p - dir * s) . n + d = 0; fp64 because dir may be almost parallel to the plane
  const V3 n{plane(0), plane(1), plane(2)};
  const V3 pd{p.x(), p.y(), p.z()};
  const V3 dir = projection_direction ? V3{projection_direction->x(), projection_direction->y(), projection_direction->z()} : n;
  const double s = (dot(pd, n) + (double)plane(3)) / dot(dir, n);
  return Point3D((float)(pd.x - dir.x * s), (float)(pd.y - dir.y * s), (float)(pd.z - dir.z * s));
}

Matrix3 EstimateHomography(const Points2D& p1, const Points2D& p2) { return homography_dlt(p1, p2); }
Matrix3 EstimateHomography(const Points2D& p1, const Points3D& p2) { return homography_dlt(p1, p2); }
Matrix3 EstimateHomography(const Points3D& p1, const Points2D& p2) { return homography_dlt(p1, p2); }
Matrix3 EstimateHomography(const Points3D& p1, const Points3D& p2) { return homography_dlt(p1, p2); }

Matrix3 EstimateKFromHomographies(const std::vector<Matrix3>& Hs) {
  // Zhang, "A flexible new technique for camera calibration", section 3.1 / appendix B
  assert(Hs.size() >= 3);
  const int n = (int)Hs.size();
  JacobiSvd svd(2 * n + 1, 6);
  auto vij = [](const Matrix3& H, int i, int j, double* v) {
    const double hi0 = H(0, i), hi1 = H(1, i), hi2 = H(2, i), hj0 = H(0, j), hj1 = H(1, j), hj2 = H(2, j);
    v[0] = hi0 * hj0; v[1] = hi0 * hj1 + hi1 * hj0; v[2] = hi1 * hj1;
    v[3] = hi2 * hj0 + hi0 * hj2; v[4] = hi2 * hj1 + hi1 * hj2; v[5] = hi2 * hj2;
  };
  for (int i = 0; i < n; ++i) {
    double v01[6], v00[6], v11[6];
    vij(Hs[i], 0, 1, v01); vij(Hs[i], 0, 0, v00); vij(Hs[i], 1, 1, v11);
    for (int c = 0; c < 6; ++c) { svd.at(2 * i, c) = v01[c]; svd.at(2 * i + 1, c) = v00[c] - v11[c]; }
  }
  svd.at(2 * n, 1) = (double)n;  // soft zero-skew constraint, weighted with the number of images
  svd.run();
  const int c = svd.smallest();
  const double B11 = svd.v(0, c), B12 = svd.v(1, c), B22 = svd.v(2, c), B13 = svd.v(3, c), B23 = svd.v(4, c), B33 = svd.v(5, c);
  const double den = B11 * B22 - B12 * B12;
  const double v0 = (B12 * B13 - B11 * B23) / den;
  const double lambda = B33 - (B13 * B13 + v0 * (B12 * B13 - B11 * B23)) / B11;
  const double alpha = std::sqrt(lambda / B11);
  const double beta = std::sqrt(lambda * B11 / den);
  const double u0 = -B13 * alpha * alpha / lambda;  // skew forced to zero
  Matrix3 K;
  K(0, 0) = (float)alpha; K(0, 2) = (float)u0;
  K(1, 1) = (float)beta;  K(1, 2) = (float)v0;
  K(2, 2) = 1.0f;
  return K;
}

std::tuple<Matrix3, Point3D> RecoverExtrinsics(const Matrix3& K_inv, const Matrix3& H) {
  // Zhang section 3.1: r0 = l K^-1 h0, r1 = l K^-1 h1, r2 = r0 x r1, t = l K^-1 h2 (float like the reference)
  auto mul = [&](int col, float scale, float* out) {
    for (int r = 0; r < 3; ++r) out[r] = scale * (K_inv(r, 0) * H(0, col) + K_inv(r, 1) * H(1, col) + K_inv(r, 2) * H(2, col));
  };
  float a0[3];
  mul(0, 1.0f, a0);
  const float l = 1.0f / std::sqrt(a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]);
  float r0[3], r1[3], t[3];
  mul(0, l, r0); mul(1, l, r1); mul(2, l, t);
  const float r2[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  Matrix3 R;
  for (int r = 0; r < 3; ++r) { R(r, 0) = r0[r]; R(r, 1) = r1[r]; R(r, 2) = r2[r]; }
  return {FixRotationMatrix(R), Point3D(t[0], t[1], t[2])};
}

Matrix3 FixRotationMatrix(const Matrix3& R) {
  double M[9], P[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r * 3 + c] = R(r, c);
  polar3(M, P);
  Matrix3 out;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out(r, c) = (float)P[r * 3 + c];
  return out;
}

Matrix3 Inverse3x3(const Matrix3& m) {
  const float c00 = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1);
  const float c01 = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2);
  const float c02 = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
  const float inv_det = 1.0f / (m(0, 0) * c00 + m(0, 1) * c01 + m(0, 2) * c02);
  Matrix3 o;
  o(0, 0) = c00 * inv_det; o(0, 1) = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * inv_det; o(0, 2) = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * inv_det;
  o(1, 0) = c01 * inv_det; o(1, 1) = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * inv_det; o(1, 2) = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * inv_det;
  o(2, 0) = c02 * inv_det; o(2, 1) = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * inv_det; o(2, 2) = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * inv_det;
  return o;
}

namespace {
// Shepperd's method as in Eigen's quaternion-from-matrix assignment; q = w x y z
void rotation_to_quaternion(const double* m /*row-major*/, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (m[7] - m[5]) * t; q[2] = (m[2] - m[6]) * t; q[3] = (m[3] - m[1]) * t;
    return;
  }
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > m[i * 4]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = std::sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
  q[1 + i] = 0.5 * t;
  t = 0.5 / t;
  q[0] = (m[k * 3 + j] - m[j * 3 + k]) * t;
  q[1 + j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
  q[1 + k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
}
}  // namespace

Quaternion QuaternionFromRotationMatrix(const Matrix3& R) {
  double m[9], q[4];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m[r * 3 + c] = R(r, c);
  rotation_to_quaternion(m, q);
  return Quaternion((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
}

void AffineToQuaternionTranslation(const Eigen::Affine3f& T, double* q, double* t) {
  // Transform::rotation() = closest rotation of the linear part (SVD); done in double here
  double M[9], R[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r * 3 + c] = T.matrix()(r, c);
  polar3(M, R);
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0) for (double& v : R) v = -v;
  rotation_to_quaternion(R, q);
  for (int i = 0; i < 3; ++i) t[i] = T.matrix()(i, 3);
}

Eigen::Affine3f QuaternionTranslationToAffine(const double* qd, const double* t) {
  // components through float, Quaterniond::normalized().toRotationMatrix().cast<float>()
  double w = (float)qd[0], x = (float)qd[1], y = (float)qd[2], z = (float)qd[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  w /= n; x /= n; y /= n; z /= n;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Eigen::Affine3f T = Eigen::Affine3f::Identity();
  auto& M = T.matrix();
  M(0, 0) = (float)(1 - (tyy + tzz)); M(0, 1) = (float)(txy - twz); M(0, 2) = (float)(txz + twy);
  M(1, 0) = (float)(txy + twz); M(1, 1) = (float)(1 - (txx + tzz)); M(1, 2) = (float)(tyz - twx);
  M(2, 0) = (float)(txz - twy); M(2, 1) = (float)(tyz + twx); M(2, 2) = (float)(1 - (txx + tyy));
  for (int i = 0; i < 3; ++i) M(i, 3) = (float)t[i];
  return T;
}

}  // namespace calibrator
