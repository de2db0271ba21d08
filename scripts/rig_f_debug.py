import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
from oracle import pyoracle as po
for (C_, F, M) in ((8, 600, 20), (8, 1100, 10), (8, 2000, 10), (4, 2000, 10), (8, 2000, 60)):
    sc = po.rig_scenario(C_, F, M)
    cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
    tq, tt = po.affine_to_qt(sc["cam_T_true"])
    args = (C_, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
    g = capi.rig_optimize(*args, options=capi.default_options(max_iterations=1000))
    o = po.rig_solve(*args, options=po.default_options(max_iterations=1000))
    print((C_, F, M), "gpu", g[5]["iterations"], g[5]["final_cost"], "oracle", o[5]["iterations"], o[5]["final_cost"],
          "cam t diff", np.abs(g[1] - o[1]).max(), "err vs planted gpu", np.abs(g[1] - tt).max(), "oracle", np.abs(o[1] - tt).max())
