"""Golden fixture of the EXTENSION problem (rig poses + shared intrinsics on pixel observations): inputs of a
3 cameras x 20 frames x 12 points case and the oracle's converged answer.
    python tests/golden/make_rigk.py    -> tests/golden/rigk_3x20x12.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import rigk_case  # noqa: E402

k = rigk_case(3, 20, 12)
mask = 1 << 8      # k3 frozen, as the helper package does for small data sets (cam_calibration.py:308)
r = po.rigk_solve(k["cams"], k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["intr0"],
                  k["cam_q0"], k["cam_t0"], k["cam_frozen"], k["frame_q0"], k["frame_t0"], const_mask=mask, huber_a=2.0,
                  options=po.default_options(max_iterations=300, function_tolerance=1e-15, gradient_tolerance=1e-13,
                                             parameter_tolerance=1e-14))
# The same problem stopped where the decisions are still ARITHMETIC, not rounding: beyond iteration ~10 of the tight solve
# the cost changes by 1e-12 on 346 (3e-15 relative, less than the rounding of the cost sum itself) while the iterate still
# moves (linear convergence, steps shrinking 5x per iteration): which of those steps is accepted depends on the order of
# the sums, so no two implementations end on the same iterate -- they all lie within the last meaningful step (~5e-7) of the
# minimiser. function_tolerance = 1e-12 ends the solve before that; THIS answer is what the HIP path is held to at 1e-11.
r12 = po.rigk_solve(k["cams"], k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["intr0"],
                    k["cam_q0"], k["cam_t0"], k["cam_frozen"], k["frame_q0"], k["frame_t0"], const_mask=mask, huber_a=2.0,
                    options=po.default_options(max_iterations=300, function_tolerance=1e-12, gradient_tolerance=1e-13,
                                               parameter_tolerance=1e-14))
log = r[6]["log"]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rigk_3x20x12.npz"),
                    cams=3, frame_offsets=k["frame_offsets"], obs_cam=k["obs_cam"], obs_world=k["obs_world"],
                    obs_uv_pix=k["obs_uv_pix"], world_xyz=k["world_xyz"], cam_frozen=k["cam_frozen"], intr0=k["intr0"],
                    cam_q0=k["cam_q0"], cam_t0=k["cam_t0"], frame_q0=k["frame_q0"], frame_t0=k["frame_t0"],
                    const_mask=mask, huber_a=2.0, intr=r[0], cam_q=r[1], cam_t=r[2], frame_q=r[3], frame_t=r[4],
                    obs_cost=r[5], final_cost=r[6]["final_cost"], initial_cost=r[6]["initial_cost"],
                    log_cost=np.array([l["cost"] for l in log]), log_cost_change=np.array([l["cost_change"] for l in log]),
                    log_step_norm=np.array([l["step_norm"] for l in log]),
                    ft12_intr=r12[0], ft12_cam_q=r12[1], ft12_cam_t=r12[2], ft12_frame_q=r12[3], ft12_frame_t=r12[4], ft12_obs_cost=r12[5],
                    ft12_final_cost=r12[6]["final_cost"], ft12_iterations=r12[6]["iterations"],
                    ft12_log_cost=np.array([l["cost"] for l in r12[6]["log"]]))
print("iterations", r[6]["iterations"], r[6]["termination"], "final cost", repr(r[6]["final_cost"]), "intr", r[0])
print("function_tolerance 1e-12: iterations", r12[6]["iterations"], r12[6]["termination"])
