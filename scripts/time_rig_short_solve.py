"""A rig solve that starts at its optimum (one LM iteration) against the full one: wall time and solver status of each.
(Round 4: the short one took 4 ms where the 54-iteration solve takes 2.3 -- what was it waiting for?)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi
C, F, M = int(os.environ.get("C", 4)), int(os.environ.get("F", 400)), int(os.environ.get("M", 300))
sc = capi.rig_scenario(C, F, M)
cq, ct = capi.affine_to_qt(sc["cam_T"]); fq, ft = capi.affine_to_qt(sc["frame_T"])
o = capi.default_options(max_iterations=1000)
for rep in range(3):
    prob = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    prob.set_state(cq, ct, fq, ft)
    t0 = time.perf_counter(); s = prob.solve(o, log_capacity=0); t1 = time.perf_counter()
    st = prob.get_state()
    print(json.dumps(dict(kind="full", iterations=s["iterations"], termination=s["termination"], solve_ms=round((t1 - t0) * 1e3, 3), status=prob.solver_status())))
    prob.close()
    prob = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    prob.set_state(*st[:4])
    t0 = time.perf_counter(); s = prob.solve(o, log_capacity=0); t1 = time.perf_counter()
    print(json.dumps(dict(kind="from the optimum", iterations=s["iterations"], termination=s["termination"], solve_ms=round((t1 - t0) * 1e3, 3), status=prob.solver_status())))
    prob.close()
