"""Where the time of k_rig_sweep_k2 goes: shader-clock cycles per phase of wave 0 of the middle workgroup, from a timing-only build
(scripts/build_variant.sh k2time cc_rig.hip --patch timing -DCC_RIG_K2_TIMING; CC_LIB_PATH=scripts/ablate_build/libcc_k2time.so). Env: C F M K."""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi, harness

Cc, F, M = int(os.environ.get("C", 8)), int(os.environ.get("F", 2000)), int(os.environ.get("M", 500))
K = os.environ.get("K", "shared")
k = harness.rigk_case(Cc, F, M, per_camera=K == "per_camera")
prob = capi.RigProblem(Cc, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["cam_frozen"], huber_a=0.0,
                       with_intrinsics=True if K == "shared" else "per_camera")
if K == "shared":
    prob.set_intrinsics(k["intr0"], 0)
else:
    for c in range(Cc):
        prob.set_camera_intrinsics(c, k["intr0"][c], 0)
prob.set_state(k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"])
rows = []
for _ in range(5):
    prob.reset()
    prob.solve(capi.default_options(max_iterations=3), log_capacity=0)
    buf = np.zeros(64)
    capi._check(capi.lib().cc_rig_debug_fetch(prob._h, b"shared_stats", buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(64)))
    rows.append(buf[40:54].copy())
prob.close()
r = np.median(np.array(rows), axis=0)
head_names = ["loads + records into LDS + barrier", "pose chain and intrinsics into scalar registers", "adjoint M", "model-cost weights + accumulators zeroed (= the rest of the head)"]
names = ["head: model-cost weights + accumulators zeroed", "evaluation + own rows (summed over passes)", "first barrier", "partner rows + accumulation",
         "second barrier + loop control", "wait for the pass's observations", "lane sums", "assembly + record store"]
wall_us = r[8] / 100.0
cyc = r[:8].sum()
print(json.dumps({"kernel": "k_rig_sweep_k2, wave 0 of the middle workgroup", "cams": Cc, "frames": F, "pts": M, "wall_us": round(float(wall_us), 2),
                  "shader_clock_mhz": round(float(cyc / max(wall_us, 1e-9)), 1), "groups_of_this_workgroup": int(r[9]), "cycles": {n: int(v) for n, v in zip(names, r[:8])},
                  "head_cycles": {n: int(v) for n, v in zip(head_names[:3], r[10:13])}}))
