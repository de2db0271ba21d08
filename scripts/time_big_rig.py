"""Wall-clock per LM iteration of the plain large-rig kernels (k_rig_elim_big / k_rig_solve_big, DESIGN.md section 4, rig
item 7) -- a measurement aid, not on the product path. One JSON line per case -> profiles/r03/rig_big.jsonl."""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import rigk_case


def time_rig(cams, frames, pts):
    sc = po.rig_scenario(cams, frames, pts)
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    prob = capi.RigProblem(cams, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    best = None
    for _ in range(5):
        prob.set_state(cq, ct, fq, ft)
        t0 = time.perf_counter()
        s = prob.solve(capi.default_options(max_iterations=40, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    prob.set_state(cq, ct, fq, ft)
    sp = prob.solve(capi.default_options(max_iterations=20, profile_kernels=1))
    per_kernel = {k: round(1e3 * v / max(1, sp["kernel_launches"][k]), 1) for k, v in sp["kernel_ms"].items() if sp["kernel_launches"].get(k)}
    prob.close()
    return dict(us_per_launch=per_kernel, problem="rig poses", cameras=cams, frames=frames, points=pts, shared_coordinates=6 * (cams - 1), iterations=s["iterations"],
                us_per_iteration=round(1e6 * best / max(1, s["iterations"]), 1), termination=s["termination"])


def time_rigk_pc(cams, frames, pts):
    k = rigk_case(cams, frames, pts, per_camera=True)
    prob = capi.RigProblem(k["cams"], k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"],
                           k["cam_frozen"], huber_a=0.0, with_intrinsics="per_camera")
    for c in range(cams):
        prob.set_camera_intrinsics(c, k["intr0"][c], 0)
    best = None
    for _ in range(5):
        for c in range(cams):
            prob.set_camera_intrinsics(c, k["intr0"][c], 0)
        prob.set_state(k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"])
        t0 = time.perf_counter()
        s = prob.solve(capi.default_options(max_iterations=40, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    prob.close()
    return dict(problem="rig poses + intrinsics per camera", cameras=cams, frames=frames, points=pts, shared_coordinates=6 * (cams - 1) + 9 * cams,
                iterations=s["iterations"], us_per_iteration=round(1e6 * best / max(1, s["iterations"]), 1), termination=s["termination"])


if __name__ == "__main__":
    for case in ((21, 30, 6), (23, 30, 6), (32, 24, 10), (40, 12, 8), (40, 2000, 100)):
        print(json.dumps(time_rig(*case)), flush=True)
    for case in ((8, 40, 30), (10, 40, 30), (17, 30, 24)):
        print(json.dumps(time_rigk_pc(*case)), flush=True)
