// mini_eigen.hh -- stand-in for the few Eigen types of the calibrator API when Eigen is not
// installed. Same names, column-major storage, float32; only the accessors the class surface and
// the pybind11 shim use. Not a linear-algebra library: the numerical work is in geometry.cpp and
// behind the C ABI.
#pragma once
#include <cmath>
#include <cstddef>
#include <memory>
#include <vector>

namespace Eigen {

template <class T>
using aligned_allocator = std::allocator<T>;

template <int N>
struct FixedVecF {
  float v[N];
  FixedVecF() { for (int i = 0; i < N; ++i) v[i] = 0.0f; }
  FixedVecF(float a, float b) { static_assert(N == 2, "2 coefficients"); v[0] = a; v[1] = b; }
  FixedVecF(float a, float b, float c) { static_assert(N == 3, "3 coefficients"); v[0] = a; v[1] = b; v[2] = c; }
  FixedVecF(float a, float b, float c, float d) { static_assert(N == 4, "4 coefficients"); v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
  static FixedVecF Zero() { return FixedVecF(); }
  static FixedVecF UnitZ() { FixedVecF r; r.v[N >= 3 ? 2 : 0] = 1.0f; return r; }
  static constexpr int size() { return N; }
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  float& x() { return v[0]; }
  float& y() { return v[1]; }
  float& z() { return v[2]; }
  float& w() { return v[3]; }
  float x() const { return v[0]; }
  float y() const { return v[1]; }
  float z() const { return v[2]; }
  float w() const { return v[3]; }
  float* data() { return v; }
  const float* data() const { return v; }
};
using Vector2f = FixedVecF<2>;
using Vector3f = FixedVecF<3>;
using Vector4f = FixedVecF<4>;

template <int N>
struct SquareMatF {  // column-major like Eigen's default
  float m[N * N];
  SquareMatF() { for (int i = 0; i < N * N; ++i) m[i] = 0.0f; }
  static SquareMatF Identity() { SquareMatF r; for (int i = 0; i < N; ++i) r.m[i * N + i] = 1.0f; return r; }
  static SquareMatF Zero() { return SquareMatF(); }
  float& operator()(int r, int c) { return m[c * N + r]; }
  float operator()(int r, int c) const { return m[c * N + r]; }
  float& operator()(int i) { return m[i]; }  // linear (column-major) index
  float operator()(int i) const { return m[i]; }
  static constexpr int rows() { return N; }
  static constexpr int cols() { return N; }
  float* data() { return m; }
  const float* data() const { return m; }
};
using Matrix3f = SquareMatF<3>;
using Matrix4f = SquareMatF<4>;

struct VectorXf {
  std::vector<float> v;
  VectorXf() {}
  explicit VectorXf(int n) : v((size_t)n, 0.0f) {}
  static VectorXf Zero(int n) { return VectorXf(n); }
  int size() const { return (int)v.size(); }
  void resize(int n) { v.assign((size_t)n, 0.0f); }
  float& operator()(int i) { return v[(size_t)i]; }
  float operator()(int i) const { return v[(size_t)i]; }
  float& operator[](int i) { return v[(size_t)i]; }
  float operator[](int i) const { return v[(size_t)i]; }
  float* data() { return v.data(); }
  const float* data() const { return v.data(); }
};

struct Quaternionf {  // coefficients stored x y z w like Eigen
  float c[4];
  Quaternionf() { c[0] = c[1] = c[2] = 0.0f; c[3] = 1.0f; }
  Quaternionf(float w, float x, float y, float z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
  float& x() { return c[0]; }
  float& y() { return c[1]; }
  float& z() { return c[2]; }
  float& w() { return c[3]; }
  float x() const { return c[0]; }
  float y() const { return c[1]; }
  float z() const { return c[2]; }
  float w() const { return c[3]; }
};

// Affine 3-D transform stored as a 4x4 matrix (Eigen::Transform<float,3,Affine>)
struct Affine3f {
  Matrix4f M;
  Affine3f() : M(Matrix4f::Identity()) {}
  explicit Affine3f(const Matrix4f& m) : M(m) {}
  static Affine3f Identity() { return Affine3f(); }
  Matrix4f& matrix() { return M; }
  const Matrix4f& matrix() const { return M; }
};

}  // namespace Eigen
