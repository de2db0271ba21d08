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

/// Plane a x + b y + c z + d = 0 through three points, normalised so that d = -1.
Plane EstimatePlaneFinite(const Point3D& p1, const Point3D& p2, const Point3D& p3);
/// Unit normal of the plane.
Point3D PlaneNormal(const Plane& plane);
/// Rotation whose third row is the plane normal (points on the plane get constant z).
Matrix3 RotationMatrixFromPlane(const Plane& plane, const Point3D& new_normal = Point3D::UnitZ());
/// Projects p onto the plane along projection_direction (plane normal if not given).
Point3D ProjectToPlane(const Plane& plane, const Point3D& p,
                       const std::optional<Point3D>& projection_direction = std::nullopt);

/// DLT homography p2 ~ H p1 (only x, y of 3-D points are used; their z must be constant).
Matrix3 EstimateHomography(const Points2D& p1, const Points2D& p2);
Matrix3 EstimateHomography(const Points2D& p1, const Points3D& p2);
Matrix3 EstimateHomography(const Points3D& p1, const Points2D& p2);
Matrix3 EstimateHomography(const Points3D& p1, const Points3D& p2);

/// Zhang's closed-form K from >= 3 world-to-image homographies (zero skew).
Matrix3 EstimateKFromHomographies(const std::vector<Matrix3>& Hs);
/// Pose (R, t) from K^-1 and a world-to-image homography.
std::tuple<Matrix3, Point3D> RecoverExtrinsics(const Matrix3& K_inv, const Matrix3& H);
/// Closest orthogonal matrix (U V^T of the SVD).
Matrix3 FixRotationMatrix(const Matrix3& R);

// ---- helpers of the class surface (not part of the reference's geometry.hh) -----------------
Matrix3 Inverse3x3(const Matrix3& m);                         // Eigen: Matrix3f::inverse()
Quaternion QuaternionFromRotationMatrix(const Matrix3& R);    // Eigen: Quaternionf(Matrix3f)
/// Affine3f -> (w x y z, t) in double: Quaterniond(T.rotation().cast<double>()), T.translation()
void AffineToQuaternionTranslation(const Eigen::Affine3f& T, double* q_wxyz, double* t_xyz);
/// (w x y z, t) -> Affine3f with the float round trip of extrinsics_calibrator.cpp:228-256
Eigen::Affine3f QuaternionTranslationToAffine(const double* q_wxyz, const double* t_xyz);

}  // namespace calibrator
