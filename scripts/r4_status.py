import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from camera_calibrator_amd import capi
for (C, F, M) in ((2, 1000, 4), (4, 400, 300), (2, 800, 4)):
    sc = capi.rig_scenario(C, F, M)
    cq, ct = capi.affine_to_qt(sc["cam_T"]); fq, ft = capi.affine_to_qt(sc["frame_T"])
    p = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    p.set_state(cq, ct, fq, ft)
    print((C, F, M), "form before", p.solver_form(), flush=True)
    t0 = time.time(); s = p.solve(capi.default_options(max_iterations=1000), log_capacity=0); dt = time.time() - t0
    print("  first solve %.3f s, status" % dt, p.solver_status(), flush=True)
    p.close()
