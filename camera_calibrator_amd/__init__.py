"""camera_calibrator_amd -- MI355X-native reprojection-error LM solver behind the
Calibrator / ExtrinsicsCalibrator surface of buq2/camera_calibrator.

Layout:
  csrc/           HIP kernels + C ABI (libcc_hip.so) + the C++ class surface + pybind11 shim
  capi.py         ctypes binding of include/cc_solver.h (what tests and bench.py call)
"""
from . import capi  # noqa: F401
