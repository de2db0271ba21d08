// binds.cpp -- pybind11 module `pycalibrator`: same class, method and keyword names as the
// reference's src/binds.cpp:47-92. Without Eigen the value types are converted by the small casters
// below (numpy arrays / sequences of floats in, numpy arrays out); with Eigen installed
// pybind11/eigen.h does it, plus the Affine3f <-> 4x4 caster the reference defines.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "calibrator.hh"
#include "extrinsics_calibrator.hh"
#include "geometry.hh"

namespace py = pybind11;

#ifdef CC_HAVE_EIGEN
#include <pybind11/eigen.h>
// With Eigen the Affine3f <-> 4x4 conversion has to be the reference's own caster (src/binds.cpp:10-45): it is
// part of the interface (only the top 3x4 is read on load) and is restated here as it stands there. Dead code in
// this image (no Eigen installed); the mini_eigen casters below are what runs.
namespace pybind11 { namespace detail {
template <typename T> struct type_caster<Eigen::Transform<T, 3, 2, 0>> {
  using Type = Eigen::Transform<T, 3, 2, 0>;
  PYBIND11_TYPE_CASTER(Type, const_name("Eigen::Transform<T,3,2,0>"));
  bool load(handle src, bool imp) {
    if (!src) return false;
    auto val = type_caster<Eigen::Matrix<T, 4, 4, 0>>();
    if (!val.load(src, imp)) return false;
    value.linear() = (*val).template block<3, 3>(0, 0);   // only the top 3x4 is read
    value.translation() = (*val).template block<3, 1>(0, 3);
    return true;
  }
  static handle cast(Type src, return_value_policy policy, handle parent) {
    auto val = type_caster<Eigen::Matrix<T, 4, 4>>();
    return val.cast(src.matrix().eval(), policy, parent);
  }
};
}}  // namespace pybind11::detail
#else
namespace pybind11 { namespace detail {

inline bool load_floats(handle src, float* out, ssize_t n) {
  if (!src) return false;
  auto arr = array_t<float, array::c_style | array::forcecast>::ensure(src);
  if (!arr) { PyErr_Clear(); return false; }
  if (arr.size() != n) return false;
  const float* p = arr.data();
  for (ssize_t i = 0; i < n; ++i) out[i] = p[i];
  return true;
}

template <int N> struct type_caster<Eigen::FixedVecF<N>> {
  using Type = Eigen::FixedVecF<N>;
  PYBIND11_TYPE_CASTER(Type, const_name("numpy.ndarray[float32[") + const_name<N>() + const_name("]]"));
  bool load(handle src, bool) { return load_floats(src, value.v, N); }
  static handle cast(const Type& src, return_value_policy, handle) {
    array_t<float> a(N);
    for (int i = 0; i < N; ++i) a.mutable_at(i) = src.v[i];
    return a.release();
  }
};

template <int N> struct type_caster<Eigen::SquareMatF<N>> {
  using Type = Eigen::SquareMatF<N>;
  PYBIND11_TYPE_CASTER(Type, const_name("numpy.ndarray[float32[") + const_name<N>() + const_name(", ") + const_name<N>() + const_name("]]"));
  bool load(handle src, bool) {
    float tmp[N * N];
    auto arr = array_t<float, array::c_style | array::forcecast>::ensure(src);
    if (!arr) { PyErr_Clear(); return false; }
    if (arr.ndim() != 2 || arr.shape(0) != N || arr.shape(1) != N) return false;
    for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) value(r, c) = arr.at(r, c);
    (void)tmp;
    return true;
  }
  static handle cast(const Type& src, return_value_policy, handle) {
    array_t<float> a({N, N});
    for (int r = 0; r < N; ++r) for (int c = 0; c < N; ++c) a.mutable_at(r, c) = src(r, c);
    return a.release();
  }
};

template <> struct type_caster<Eigen::VectorXf> {
  PYBIND11_TYPE_CASTER(Eigen::VectorXf, const_name("numpy.ndarray[float32[n]]"));
  bool load(handle src, bool) {
    auto arr = array_t<float, array::c_style | array::forcecast>::ensure(src);
    if (!arr) { PyErr_Clear(); return false; }
    value.resize((int)arr.size());
    const float* p = arr.data();
    for (ssize_t i = 0; i < arr.size(); ++i) value(static_cast<int>(i)) = p[i];
    return true;
  }
  static handle cast(const Eigen::VectorXf& src, return_value_policy, handle) {
    array_t<float> a(src.size());
    for (int i = 0; i < src.size(); ++i) a.mutable_at(i) = src(i);
    return a.release();
  }
};

template <> struct type_caster<Eigen::Affine3f> {
  PYBIND11_TYPE_CASTER(Eigen::Affine3f, const_name("numpy.ndarray[float32[4, 4]]"));
  bool load(handle src, bool) {
    auto arr = array_t<float, array::c_style | array::forcecast>::ensure(src);
    if (!arr) { PyErr_Clear(); return false; }
    if (arr.ndim() != 2 || arr.shape(0) != 4 || arr.shape(1) != 4) return false;
    value = Eigen::Affine3f::Identity();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) value.matrix()(r, c) = arr.at(r, c);  // top 3x4 only
    return true;
  }
  static handle cast(const Eigen::Affine3f& src, return_value_policy, handle) {
    array_t<float> a({4, 4});
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) a.mutable_at(r, c) = src.matrix()(r, c);
    return a.release();
  }
};

template <> struct type_caster<Eigen::Quaternionf> {  // w x y z (extension: lets Optimize be called from Python)
  PYBIND11_TYPE_CASTER(Eigen::Quaternionf, const_name("numpy.ndarray[float32[4]]"));
  bool load(handle src, bool) {
    float q[4];
    if (!load_floats(src, q, 4)) return false;
    value = Eigen::Quaternionf(q[0], q[1], q[2], q[3]);
    return true;
  }
  static handle cast(const Eigen::Quaternionf& src, return_value_policy, handle) {
    array_t<float> a(4);
    a.mutable_at(0) = src.w(); a.mutable_at(1) = src.x(); a.mutable_at(2) = src.y(); a.mutable_at(3) = src.z();
    return a.release();
  }
};

}}  // namespace pybind11::detail
#endif

PYBIND11_MODULE(pycalibrator, m) {
  m.doc() = "Camera calibrator (MI355X-native solver behind the buq2/camera_calibrator surface)";
  using calibrator::Calibrator;
  using calibrator::ExtrinsicsCalibrator;

  py::class_<Calibrator>(m, "Calibrator")
      .def(py::init<int, int>())
      .def("EstimateOpenCv", &Calibrator::EstimateOpenCv, py::arg("img_points"), py::arg("world_points"))
      .def("Estimate", &Calibrator::Estimate, py::arg("img_points"), py::arg("world_points"))
      .def("Optimize", &Calibrator::Optimize)
      .def("GetK", &Calibrator::GetK)
      .def("SetK", &Calibrator::SetK, py::arg("K"))
      .def("GetDistortion", &Calibrator::GetDistortion)
      .def("SetDistortion", &Calibrator::SetDistortion, py::arg("distortion"))
      .def("ForceDistortionToConstant", &Calibrator::ForceDistortionToConstant)
      .def("Undistort", &Calibrator::Undistort, py::arg("img_points"))
      .def("Distort", &Calibrator::Distort, py::arg("normalized_points"))
      // additions of this build
      .def("SetDevice", &Calibrator::SetDevice, py::arg("device"))
      .def("SetDevices", &Calibrator::SetDevices, py::arg("devices"))
      .def("LastStatus", &Calibrator::LastStatus)
      .def("LastIterations", &Calibrator::LastIterations)
      .def("LastSolverReruns", &Calibrator::LastSolverReruns)
      .def("LastSolverForm", &Calibrator::LastSolverForm)
      .def("LastSolverNote", &Calibrator::LastSolverNote)
      .def("LastFinalCost", &Calibrator::LastFinalCost);

  py::class_<ExtrinsicsCalibrator>(m, "ExtrinsicsCalibrator")
      .def(py::init<>())
      .def("AddCameraTRig", &ExtrinsicsCalibrator::AddCameraTRig, py::arg("camera_T_rig"), py::arg("freeze") = false)
      .def("GetCameraTRig", &ExtrinsicsCalibrator::GetCameraTRig, py::arg("id"))
      .def("AddObservationFrame", &ExtrinsicsCalibrator::AddObservationFrame, py::arg("rig_T_world"))
      .def("GetObservationFrame", &ExtrinsicsCalibrator::GetObservationFrame, py::arg("id"))
      .def("AddWorldPoint", &ExtrinsicsCalibrator::AddWorldPoint, py::arg("frame_id"), py::arg("world_point"))
      .def("AddObservation", &ExtrinsicsCalibrator::AddObservation, py::arg("camera_id"), py::arg("world_point_id"), py::arg("image_point"))
      .def("Optimize", &ExtrinsicsCalibrator::Optimize)
      .def("Serialize", &ExtrinsicsCalibrator::Serialize, py::arg("fname"))
      .def("Parse", &ExtrinsicsCalibrator::Parse, py::arg("fname"))
      .def("RemoveObservationFrame", &ExtrinsicsCalibrator::RemoveObservationFrame, py::arg("observation_frame_id"))
      .def("RemoveObservationFrames", &ExtrinsicsCalibrator::RemoveObservationFrames, py::arg("observation_frame_ids"))
      // additions of this build
      .def("SetDevice", &ExtrinsicsCalibrator::SetDevice, py::arg("device"))
      .def("SetDevices", &ExtrinsicsCalibrator::SetDevices, py::arg("devices"))
      .def("SetVerbose", &ExtrinsicsCalibrator::SetVerbose, py::arg("verbose"))
      .def("LastStatus", &ExtrinsicsCalibrator::LastStatus)
      .def("LastIterations", &ExtrinsicsCalibrator::LastIterations)
      .def("LastSolverReruns", &ExtrinsicsCalibrator::LastSolverReruns)
      .def("LastSolverForm", &ExtrinsicsCalibrator::LastSolverForm)
      .def("LastSolverNote", &ExtrinsicsCalibrator::LastSolverNote)
      .def("LastFinalCost", &ExtrinsicsCalibrator::LastFinalCost)
      .def("NumCameras", &ExtrinsicsCalibrator::NumCameras)
      .def("NumObservationFrames", &ExtrinsicsCalibrator::NumObservationFrames)
      .def("NumWorldPoints", &ExtrinsicsCalibrator::NumWorldPoints)
      .def("IsCameraFrozen", &ExtrinsicsCalibrator::IsCameraFrozen, py::arg("id"))
      .def("NumObservations", &ExtrinsicsCalibrator::NumObservations, py::arg("frame_id"))
      .def("GetObservation", [](const ExtrinsicsCalibrator& self, size_t frame_id, size_t k) {
        size_t cam, idx, id; calibrator::Point2D p; double cost;
        self.GetObservation(frame_id, k, &cam, &idx, &id, &p, &cost);
        return py::make_tuple(cam, idx, id, p, cost);
      }, py::arg("frame_id"), py::arg("k"));

  m.def("EstimatePlaneFinite", &calibrator::EstimatePlaneFinite, py::arg("p1"), py::arg("p2"), py::arg("p3"));
  m.def("PlaneNormal", &calibrator::PlaneNormal, py::arg("plane"));
  m.def("RotationMatrixFromPlane", &calibrator::RotationMatrixFromPlane, py::arg("plane"),
        py::arg("new_normal") = calibrator::Point3D::UnitZ());
  m.def("ProjectToPlane", &calibrator::ProjectToPlane, py::arg("plane"), py::arg("p"), py::arg("projection_direction") = std::nullopt);
  m.def("EstimateHomography", py::overload_cast<const calibrator::Points2D&, const calibrator::Points2D&>(&calibrator::EstimateHomography),
        "Estimate homography", py::arg("p1"), py::arg("p2"));
  m.def("EstimateKFromHomographies", &calibrator::EstimateKFromHomographies, py::arg("Hs"));
  m.def("RecoverExtrinsics", &calibrator::RecoverExtrinsics, "Recover extrinsics from inverted calibration matrix and homography",
        py::arg("K_inv"), py::arg("H"));
  m.def("FixRotationMatrix", &calibrator::FixRotationMatrix, py::arg("R"));
}
