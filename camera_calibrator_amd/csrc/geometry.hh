// geometry.hh -- plane helpers and Zhang's closed-form initialisation (host side, fp64 inside).
// Same free functions as the reference's src/geometry.hh:13-48; they feed Calibrator::Estimate
// with the initial K and poses (src/calibrator.cpp:47-66).  The SVDs the reference takes from
// Eigen::JacobiSVD are done with a one-sided Jacobi (Hestenes) SVD here.
#pragma once
#include <optional>
#include <tuple>
#include <vector>

#include "types.hh"

namespace calibrator {

// ---- homographies and Zhang's closed form ------------------------------------------------------
/// DLT homography with `to` ~ H * `from` (of 3-D points only x and y are used: their z must be constant).
Matrix3 EstimateHomography(const Points2D& from, const Points2D& to);
Matrix3 EstimateHomography(const Points2D& from, const Points3D& to);
Matrix3 EstimateHomography(const Points3D& from, const Points2D& to);
Matrix3 EstimateHomography(const Points3D& from, const Points3D& to);
/// Zhang's closed-form K (zero skew) from at least three world-to-image homographies.
Matrix3 EstimateKFromHomographies(const std::vector<Matrix3>& homographies);
/// Pose (R, t) of the board from the inverse camera matrix and its world-to-image homography.
std::tuple<Matrix3, Point3D> RecoverExtrinsics(const Matrix3& K_inverse, const Matrix3& homography);
/// Closest orthogonal matrix to a nearly-orthogonal one (U V^T of its SVD).
Matrix3 FixRotationMatrix(const Matrix3& nearly_orthogonal);

// ---- planes ------------------------------------------------------------------------------------------
/// Plane through three points, scaled so that its last coefficient is -1.
Plane EstimatePlaneFinite(const Point3D& a, const Point3D& b, const Point3D& c);
/// Unit normal of a plane.
Point3D PlaneNormal(const Plane& plane);
/// Rotation that turns the plane normal into `target_normal` (points of the plane get a constant z for UnitZ).
Matrix3 RotationMatrixFromPlane(const Plane& plane, const Point3D& target_normal = Point3D::UnitZ());
/// Point where the line through `point` along `direction` (default: the plane normal) meets the plane.
Point3D ProjectToPlane(const Plane& plane, const Point3D& point,
                       const std::optional<Point3D>& direction = std::nullopt);

// ---- helpers of the class surface (not part of the reference's geometry.hh) -----------------
Matrix3 Inverse3x3(const Matrix3& m);                         // Eigen: Matrix3f::inverse()
Quaternion QuaternionFromRotationMatrix(const Matrix3& R);    // Eigen: Quaternionf(Matrix3f)
/// Affine3f -> (w x y z, t) in double: Quaterniond(T.rotation().cast<double>()), T.translation()
void AffineToQuaternionTranslation(const Eigen::Affine3f& T, double* q_wxyz, double* t_xyz);
/// (w x y z, t) -> Affine3f with the float round trip of extrinsics_calibrator.cpp:228-256
Eigen::Affine3f QuaternionTranslationToAffine(const double* q_wxyz, const double* t_xyz);

}  // namespace calibrator
