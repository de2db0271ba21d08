"""Golden result of the oracle on BASELINE.json configs[4] (rig, 8 cameras x 2000 frames x 500 points,
8M observations; the scenario of test_extrinsics_calibrator.cpp:48-134 at that size). The oracle needs
about a minute for it, which is too long for the test-suite, so its answer is committed here:
    python tests/golden/make_rig_c5.py      -> tests/golden/rig_c5_oracle.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po  # noqa: E402

C_, F, M = 8, 2000, 500
sc = po.rig_scenario(C_, F, M)
cq, ct = po.affine_to_qt(sc["cam_T"])
fq, ft = po.affine_to_qt(sc["frame_T"])
o = po.rig_solve(C_, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct,
                 sc["cam_frozen"], fq, ft, options=po.default_options(max_iterations=1000))
pick = np.r_[0:8, F // 2:F // 2 + 8, F - 8:F]
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rig_c5_oracle.npz"),
         cams=C_, frames=F, pts=M, iterations=o[5]["iterations"], termination=o[5]["termination"],
         initial_cost=o[5]["initial_cost"], final_cost=o[5]["final_cost"],
         costs=np.array([l["cost"] for l in o[5]["log"]]), accepted=np.array([l["accepted"] for l in o[5]["log"]]),
         cam_q=o[0], cam_t=o[1], frame_pick=pick, frame_q=o[2][pick], frame_t=o[3][pick],
         obs_cost_sum=o[4].sum(), obs_cost_head=o[4][:64])
print("iterations", o[5]["iterations"], "final cost", repr(o[5]["final_cost"]))
