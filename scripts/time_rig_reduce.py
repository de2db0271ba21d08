"""Where the time of the rig path's reduce + solve + update launch goes: wall-clock marks left by the solving block of
a timing-only build (scripts/build_variant.sh rigtime cc_rig.hip --patch timing -DCC_RIG_TIMING; CC_LIB_PATH=scripts/ablate_build/libcc_rigtime.so).
Env: C F M (cameras, frames, points per frame). Prints the stage durations in microseconds (last iteration of a solve)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi

Cc, F, M = int(os.environ.get("C", 4)), int(os.environ.get("F", 400)), int(os.environ.get("M", 300))
K = os.environ.get("K", "none")     # none | shared | per_camera (the intrinsics extension)
if K == "none":
    sc = capi.rig_scenario(Cc, F, M)
    cq, ct = capi.affine_to_qt(sc["cam_T"]); fq, ft = capi.affine_to_qt(sc["frame_T"])
    prob = capi.RigProblem(Cc, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
else:
    from camera_calibrator_amd import harness
    k = harness.rigk_case(Cc, F, M, per_camera=K == "per_camera")
    cq, ct, fq, ft = k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"]
    prob = capi.RigProblem(Cc, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["cam_frozen"], huber_a=0.0,
                           with_intrinsics=True if K == "shared" else "per_camera")
    if K == "shared":
        prob.set_intrinsics(k["intr0"], 0)
    else:
        for c in range(Cc):
            prob.set_camera_intrinsics(c, k["intr0"][c], 0)
prob.set_state(cq, ct, fq, ft)
rows, erows, swrows = [], [], []
for _ in range(5):
    prob.reset()
    s = prob.solve(capi.default_options(max_iterations=12), log_capacity=0)
    buf = np.zeros(64)
    capi._check(capi.lib().cc_rig_debug_fetch(prob._h, b"shared_stats", buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(64)))
    b4 = buf[48:53] / 100.0
    sclk_mhz = buf[54] / max(buf[55], 1.0) * 100.0
    t = buf[8:16] / 100.0   # 100 MHz ticks -> us
    rows.append(np.concatenate([np.diff(t), [buf[16] / 100.0, buf[17] / 100.0, (buf[19] - buf[18]) / 100.0]]))
    e = buf[20:30] / 100.0
    erows.append(np.diff(e))
    swrows.append(np.diff(buf[33:38] / 100.0))   # (mark 0 belongs to the solve's last, empty launch)
prob.close()
d = np.median(np.array(rows), axis=0)
ed = np.median(np.array(erows), axis=0)
enames = ["control block read", "statistics + decision", "lane tables", "frame loads + broadcast (first frames)", "6x6 Cholesky",
          "columns z, y + staging", "Schur products (MFMA)", "remaining frame passes", "partial row store"]
names = ["column sums", "drain + arrival", "reads + assembly", "rhs/diagonal/tests", "factorisation + substitutions",
         "candidates + control block", "flag + pose update", "(of the factorisation: panels on wave 0)", "(trailing updates)",
         "(backward substitution + step store)"]
swd = np.median(np.array(swrows), axis=0)
print(json.dumps({"kernel": "k_rig_sweep (middle workgroup, from the records barrier on)", "cams": Cc, "frames": F, "pts": M, "total_us": float(swd.sum()),
                  **{n: round(float(v), 2) for n, v in zip(["model-cost term + rotation setup", "first pass", "remaining passes", "cross-wave reduction + block store"], swd)}}))
print(json.dumps({"kernel": "k_rig_elim (block 0)", "cams": Cc, "frames": F, "pts": M, "total_us": float(ed.sum()),
                  **{n: round(float(v), 2) for n, v in zip(enames, ed)}}))
if b4.sum() > 0:
    print(json.dumps({"kernel": "chol_block4 (thread 0, summed over the blocks of four columns, last solve)", **{n: round(float(v), 2) for n, v in zip(
        ["LDS reads of the diagonal block + row", "4 x 4 factor + row solve", "row store + barrier", "trailing update (MFMA)", "barrier"], b4)},
        "shader_clock_mhz_during_the_factorisation": round(float(sclk_mhz), 1)}))
print(json.dumps({"kernel": "k_rig_reduce (solving block)", "cams": Cc, "frames": F, "pts": M, "intrinsics": K, "total_us": float(d[:7].sum()), **{n: round(float(v), 2) for n, v in zip(names, d)}}))
