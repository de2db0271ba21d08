// types.hh -- the float32 value types of the calibrator API (reference: src/types.hh:10-20).
//
// The reference takes these from Eigen. Eigen is not part of this build environment, so when
// <Eigen/Dense> is absent a minimal stand-in with the same names, storage order and the handful of
// accessors the class surface needs is provided (mini_eigen.hh). With Eigen installed the real
// types are used and the classes compile unchanged.
#pragma once
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(CC_FORCE_MINI_EIGEN)
#define CC_HAVE_EIGEN 1
#endif
#endif

#ifdef CC_HAVE_EIGEN
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <Eigen/StdVector>
#else
#include "mini_eigen.hh"
#endif

namespace calibrator {

using Point2D = Eigen::Vector2f;
using Point3D = Eigen::Vector3f;
using Vector3 = Point3D;
using Vector4 = Eigen::Vector4f;
using Plane = Eigen::Vector4f;
using Matrix3 = Eigen::Matrix3f;
using Matrix4 = Eigen::Matrix4f;
using DynamicVector = Eigen::VectorXf;
using Quaternion = Eigen::Quaternionf;
using Points2D = std::vector<Point2D, Eigen::aligned_allocator<Point2D>>;
using Points3D = std::vector<Point3D, Eigen::aligned_allocator<Point3D>>;

}  // namespace calibrator
