"""Wall time of cc_rig_optimize -- what ExtrinsicsCalibrator::Optimize calls: regrouping on the host, upload, records, solve,
per-observation costs, read-back, teardown -- against the solve alone (CC_RIG_HOST_TIMING=1 prints the phases to stderr).
C / F / M from the environment as scripts/bench_rig.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi
C, F, M = int(os.environ.get("C", 4)), int(os.environ.get("F", 400)), int(os.environ.get("M", 300))
sc = capi.rig_scenario(C, F, M)
cq, ct = capi.affine_to_qt(sc["cam_T"]); fq, ft = capi.affine_to_qt(sc["frame_T"])
o = capi.default_options(max_iterations=1000)
ts = []
for rep in range(int(os.environ.get("REPS", 6))):
    t0 = time.perf_counter()
    r = capi.rig_optimize(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft, options=o, log_capacity=0)
    ts.append(time.perf_counter() - t0)
s = r[5]
print(json.dumps(dict(call="cc_rig_optimize", cams=C, frames=F, pts=M, observations=len(sc["obs_cam"]), iterations=s["iterations"],
                      first_call_ms=round(ts[0] * 1e3, 3), wall_ms_median=round(float(np.median(ts[1:])) * 1e3, 3), final_cost=s["final_cost"])))
