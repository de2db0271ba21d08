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

// single points and small fixed-size linear algebra (all float32, as the reference's API stores them)
typedef Eigen::Vector2f Point2D;
typedef Eigen::Vector3f Point3D;
typedef Point3D Vector3;
typedef Eigen::Vector4f Vector4;
typedef Eigen::Vector4f Plane;          ///< (a, b, c, d) of a x + b y + c z + d = 0
typedef Eigen::Matrix3f Matrix3;
typedef Eigen::Matrix4f Matrix4;
typedef Eigen::Quaternionf Quaternion;
typedef Eigen::VectorXf DynamicVector;  ///< distortion coefficients k1 k2 p1 p2 k3

// point lists of one view
typedef std::vector<Point2D, Eigen::aligned_allocator<Point2D>> Points2D;
typedef std::vector<Point3D, Eigen::aligned_allocator<Point3D>> Points3D;

}  // namespace calibrator
