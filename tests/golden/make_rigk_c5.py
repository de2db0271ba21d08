"""Golden result of the oracle on BASELINE.json configs[4] WITH shared intrinsics (EXTENSION, cc_rigk_*: 8 cameras
x 2000 frames x 500 points, 8M pixel observations, tests/helpers.py rigk_case). The oracle needs minutes for it,
which is too long for the test-suite, so its answer is committed here:
    python tests/golden/make_rigk_c5.py      -> tests/golden/rigk_c5_oracle.npz
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import rigk_case  # noqa: E402

C_, F, M = 8, 2000, 500
k = rigk_case(C_, F, M)
t0 = time.time()
o = po.rigk_solve(k["cams"], k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["intr0"],
                  k["cam_q0"], k["cam_t0"], k["cam_frozen"], k["frame_q0"], k["frame_t0"], const_mask=0, huber_a=0.0,
                  options=po.default_options(max_iterations=200, num_threads=8))
print("oracle seconds", time.time() - t0)
pick = np.r_[0:8, F // 2:F // 2 + 8, F - 8:F]
s = o[6]
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rigk_c5_oracle.npz"),
         cams=C_, frames=F, pts=M, iterations=s["iterations"], termination=s["termination"],
         initial_cost=s["initial_cost"], final_cost=s["final_cost"],
         costs=np.array([l["cost"] for l in s["log"]]), accepted=np.array([l["accepted"] for l in s["log"]]),
         intr=o[0], cam_q=o[1], cam_t=o[2], frame_pick=pick, frame_q=o[3][pick], frame_t=o[4][pick],
         obs_cost_sum=o[5].sum(), obs_cost_head=o[5][:64])
print("iterations", s["iterations"], s["termination"], "final cost", repr(s["final_cost"]), "intr", o[0])
