"""One-shot rig solve (cc_rig_optimize: create + set_state + solve + get_state + destroy), C4 size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
from camera_calibrator_amd import capi
from oracle import pyoracle as po
sc = po.rig_scenario(4, 400, 300)
cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
for rep in range(4):
    t0 = time.perf_counter()
    r = capi.rig_optimize(4, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
    print("one-shot rig %.3f ms (%d iterations)" % ((time.perf_counter() - t0) * 1e3, r[5]["iterations"]))
for rep in range(3):
    t = [time.perf_counter()]
    p = capi.RigProblem(4, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"]); t.append(time.perf_counter())
    p.set_state(cq, ct, fq, ft); t.append(time.perf_counter())
    s = p.solve(capi.default_options(max_iterations=1000, use_graph=0), log_capacity=0); t.append(time.perf_counter())
    p.get_state(); t.append(time.perf_counter())
    p.close(); t.append(time.perf_counter())
    print("create %.3f set_state %.3f solve %.3f get_state %.3f destroy %.3f ms" % tuple((b - a) * 1e3 for a, b in zip(t, t[1:])))
