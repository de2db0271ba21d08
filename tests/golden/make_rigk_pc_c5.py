"""Golden result of the oracle on BASELINE.json configs[4] AS WORDED ("full intrinsics+extrinsics co-optimisation"):
8 cameras x 2000 frames x 500 points, 8M pixel observations, one set of 9 intrinsics PER CAMERA (EXTENSION,
cc_rigk_create_per_camera; scenario tests/helpers.py rigk_case(per_camera=True); 6*7 + 9*8 = 114 shared coordinates).
The oracle needs many minutes for it (a dense 114-wide reduction over 8M observations per iteration), which is too
long for the test-suite, so its answer is committed here:
    python tests/golden/make_rigk_pc_c5.py      -> tests/golden/rigk_pc_c5_oracle.npz
The fixture is the ORACLE's output (a regression pin and a fixed target for the HIP path), not the reference's: the
reference has no such problem."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import rigk_case  # noqa: E402

C_, F, M = 8, 2000, 500
k = rigk_case(C_, F, M, per_camera=True)
t0 = time.time()
o = po.rigk_solve_per_camera(k["cams"], k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"],
                             k["intr0"], k["cam_q0"], k["cam_t0"], k["cam_frozen"], k["frame_q0"], k["frame_t0"],
                             const_masks=None, huber_a=0.0,
                             options=po.default_options(max_iterations=200, num_threads=8))
print("oracle seconds", time.time() - t0)
pick = np.r_[0:8, F // 2:F // 2 + 8, F - 8:F]
s = o[6]
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rigk_pc_c5_oracle.npz"),
         cams=C_, frames=F, pts=M, iterations=s["iterations"], termination=s["termination"],
         initial_cost=s["initial_cost"], final_cost=s["final_cost"],
         costs=np.array([l["cost"] for l in s["log"]]), accepted=np.array([l["accepted"] for l in s["log"]]),
         intr=o[0], cam_q=o[1], cam_t=o[2], frame_pick=pick, frame_q=o[3][pick], frame_t=o[4][pick],
         obs_cost_sum=o[5].sum(), obs_cost_head=o[5][:64])
print("iterations", s["iterations"], s["termination"], "final cost", repr(s["final_cost"]), "intr", o[0])
