import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
from oracle import pyoracle as po
C_, F, M = 8, int(os.environ.get("F", 2000)), int(os.environ.get("M", 500))
sc = po.rig_scenario(C_, F, M)
cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
tq, tt = po.affine_to_qt(sc["cam_T_true"])
prob = capi.RigProblem(C_, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
prob.set_state(cq, ct, fq, ft)
for kw in (dict(), dict(function_tolerance=1e-12, max_iterations=400)):
    prob.reset()
    s = prob.solve(capi.default_options(max_iterations=1000, **kw) if "max_iterations" not in kw else capi.default_options(**kw))
    r = prob.get_state()
    print(kw, s["termination"], s["iterations"], s["initial_cost"], s["final_cost"])
    print(" t err per cam", np.abs(r[1] - tt).max(axis=1))
    print(" init err     ", np.abs(ct - tt).max(axis=1))
    print(" last costs", [round(l["cost"], 3) for l in s["log"][-6:]])
# cost at the planted rig with the found frame poses
prob.set_state(tq, tt, r[2], r[3])
print("cost at planted cams + found frames", prob.eval())
