"""Calibrator::Estimate at BASELINE configs[2] size through the C ABI: cc_zhang_init + cc_intrinsics_optimize (two uploads)
against cc_intrinsics_estimate (one). Wall time of the calls, host arrays in pageable memory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
off, uv, xyz = capi.make_intrinsics_problem(1000, 500)
def two():
    K0, q0, t0 = capi.zhang_init(off, uv, xyz)
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    return capi.intrinsics_optimize(off, uv, xyz, intr0, q0.astype(np.float64), t0.astype(np.float64), log_capacity=0)
def one():
    return capi.intrinsics_estimate(off, uv, xyz, log_capacity=0)
for f in (two, one, two, one):
    f(); f()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    print(f.__name__, "median ms", round(float(np.median(ts)) * 1e3, 3))
