"""Rig (ExtrinsicsCalibrator) path on one GPU: C cameras x F frames x M points of the reference's rig test scenario
(include/cc_harness.h cc_rig_scenario), every camera sees every point. Repeated complete solves, GPU timings only
(parity is the test-suite's job); meant to be run under rocprofv3 for the per-kernel split. Not part of the driver
contract (bench.py is, and it reports the same configurations under `configs`).
    C=4 F=400 M=300 python scripts/bench_rig.py            # BASELINE.json configs[3] size (default)
    C=8 F=2000 M=500 K=shared python scripts/bench_rig.py  # configs[4] size with the shared-intrinsics extension
    K=per_camera ...                                       # one camera model per camera"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (loads the ROCm runtime the library binds to)
from camera_calibrator_amd import capi

C, F, M = int(os.environ.get("C", 4)), int(os.environ.get("F", 400)), int(os.environ.get("M", 300))
K = os.environ.get("K", "none")
REPS = int(os.environ.get("REPS", 5))
if K == "none":
    sc = capi.rig_scenario(C, F, M)
    cq, ct = capi.affine_to_qt(sc["cam_T"])
    fq, ft = capi.affine_to_qt(sc["frame_T"])
    n_obs = len(sc["obs_cam"])
    prob = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
else:
    from camera_calibrator_amd import harness
    k = harness.rigk_case(C, F, M, per_camera=K == "per_camera")
    cq, ct, fq, ft = k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"]
    n_obs = len(k["obs_cam"])
    prob = capi.RigProblem(C, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["cam_frozen"], huber_a=0.0,
                           with_intrinsics=True if K == "shared" else "per_camera")
    if K == "shared":
        prob.set_intrinsics(k["intr0"], 0)
    else:
        for c in range(C):
            prob.set_camera_intrinsics(c, k["intr0"][c], 0)
prob.set_state(cq, ct, fq, ft)
o = capi.default_options(max_iterations=1000, use_graph=int(os.environ.get("GRAPH", 1)), check_interval=int(os.environ.get("CHECK", 4)))
s = prob.solve(o, log_capacity=0)
ts = []
for _ in range(REPS):
    prob.reset()
    t0 = time.perf_counter()
    s = prob.solve(o, log_capacity=0)
    ts.append(time.perf_counter() - t0)
per_kernel = None
if os.environ.get("PROFILE"):   # one more solve with hipEvents around every launch: mean over the launches that did work (us)
    prob.reset()
    sp = prob.solve(capi.default_options(max_iterations=1000, profile_kernels=1), log_capacity=0)
    per_kernel = {k: round(1e3 * v / sp["kernel_launches"][k], 1) for k, v in sp["kernel_ms"].items() if sp["kernel_launches"].get(k)}
print(json.dumps(dict(cams=C, frames=F, pts=M, intrinsics=K, kernel_us_per_full_launch=per_kernel, observations=n_obs, iterations=s["iterations"], termination=s["termination"],
                      gpu_solve_ms=float(np.median(ts) * 1e3), gpu_us_per_iteration=float(np.median(ts) * 1e6 / max(1, s["iterations"])),
                      gpu_residuals_per_s=2.0 * n_obs * s["iterations"] / float(np.median(ts)), final_cost=s["final_cost"])))
prob.close()
