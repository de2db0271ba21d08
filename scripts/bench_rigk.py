"""EXTENSION timing: rig poses + 9 shared intrinsics on pixel observations at the sizes of
BASELINE.json configs[3] (4 cameras x 400 frames x 300 points) and, GPU only, configs[4] (8 x 2000 x 500).
Not part of the driver contract; numbers are quoted in DESIGN.md."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import RIGK_INTR_TRUE, rigk_case


def run(cams, frames, pts, cpu=True):
    k = rigk_case(cams, frames, pts)
    prob = capi.RigProblem(cams, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["cam_frozen"],
                           huber_a=0.0, with_intrinsics=True)
    prob.set_intrinsics(k["intr0"], 0)
    prob.set_state(k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"])
    s = prob.solve()
    ts = []
    for _ in range(3):
        prob.reset(); t0 = time.perf_counter(); s = prob.solve(); ts.append(time.perf_counter() - t0)
    intr = prob.get_intrinsics()
    out = dict(config=f"rig+intrinsics {cams} cams x {frames} frames x {pts} pts", observations=len(k["obs_cam"]), iterations=s["iterations"],
               termination=s["termination"], gpu_solve_ms=float(np.median(ts) * 1e3), gpu_us_per_iteration=float(np.median(ts) * 1e6 / max(1, s["iterations"])),
               final_cost=s["final_cost"], focal_error=float(np.abs(intr[:2] / RIGK_INTR_TRUE[:2] - 1).max()))
    if cpu:
        t0 = time.perf_counter()
        r = po.rigk_solve(cams, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["intr0"], k["cam_q0"], k["cam_t0"],
                          k["cam_frozen"], k["frame_q0"], k["frame_t0"])
        out.update(cpu_solve_ms=(time.perf_counter() - t0) * 1e3, cpu_iterations=r[6]["iterations"], cpu_final_cost=r[6]["final_cost"],
                   intr_max_rel_diff=float(np.abs(intr / r[0] - 1)[:4].max()))
    prob.close()
    print(json.dumps(out), flush=True)


run(4, 400, 300)
run(8, 2000, 500, cpu=False)
