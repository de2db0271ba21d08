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
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rigk_3x20x12.npz"),
                    cams=3, frame_offsets=k["frame_offsets"], obs_cam=k["obs_cam"], obs_world=k["obs_world"],
                    obs_uv_pix=k["obs_uv_pix"], world_xyz=k["world_xyz"], cam_frozen=k["cam_frozen"], intr0=k["intr0"],
                    cam_q0=k["cam_q0"], cam_t0=k["cam_t0"], frame_q0=k["frame_q0"], frame_t0=k["frame_t0"],
                    const_mask=mask, huber_a=2.0, intr=r[0], cam_q=r[1], cam_t=r[2], frame_q=r[3], frame_t=r[4],
                    obs_cost=r[5], final_cost=r[6]["final_cost"], initial_cost=r[6]["initial_cost"])
print("iterations", r[6]["iterations"], r[6]["termination"], "final cost", repr(r[6]["final_cost"]), "intr", r[0])
