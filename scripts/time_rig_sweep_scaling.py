"""How the poses-only rig sweep's launch time scales with the observations per group (8 cameras x 2000 frames x M points):
per-launch microseconds from profiled solves (hipEvents around every launch), frame form and group form."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi
C, F = int(os.environ.get("C", 8)), int(os.environ.get("F", 2000))
for M in (64, 125, 250, 500, 1000):
    sc = capi.rig_scenario(C, F, M)
    cq, ct = capi.affine_to_qt(sc["cam_T"]); fq, ft = capi.affine_to_qt(sc["frame_T"])
    prob = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    prob.set_state(cq, ct, fq, ft)
    prob.solve(capi.default_options(max_iterations=10), log_capacity=0)
    prob.reset()
    p = prob.solve(capi.default_options(max_iterations=20, profile_kernels=1), log_capacity=0)
    prob.close()
    per = {k: round(1e3 * v / max(1, p["kernel_launches"][k]), 2) for k, v in p["kernel_ms"].items() if p["kernel_launches"].get(k)}
    print(json.dumps({"form": "group" if os.environ.get("CC_RIG_SWEEP_FRAME") == "0" else "frame", "cams": C, "frames": F, "pts": M, "observations": len(sc["obs_cam"]),
                      "us_per_launch": per, "sweep_ns_per_observation": round(1e3 * per["sweep"] / len(sc["obs_cam"]), 4)}))
