"""Rig (ExtrinsicsCalibrator) path on one GPU at BASELINE.json configs[3] size: 4 cameras, 400 frames x
300 points, every camera sees every point (480,000 observations). Prints GPU vs oracle timings and
checks parity. Not part of the driver contract (bench.py is); numbers are quoted in DESIGN.md."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
from oracle import pyoracle as po

C, F, M = int(os.environ.get("C", 4)), int(os.environ.get("F", 400)), int(os.environ.get("M", 300))
sc = po.rig_scenario(C, F, M)
cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
prob = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
prob.set_state(cq, ct, fq, ft)
s = prob.solve()
ts = []
for _ in range(5):
    prob.reset()
    t0 = time.perf_counter(); s = prob.solve(); ts.append(time.perf_counter() - t0)
g = prob.get_state()
t0 = time.perf_counter()
o = po.rig_solve(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
t_cpu = time.perf_counter() - t0
n_obs = len(sc["obs_cam"])
out = dict(cams=C, frames=F, pts=M, observations=n_obs, iterations=s["iterations"], termination=s["termination"],
           gpu_solve_ms=float(np.median(ts) * 1e3), gpu_ms_per_iteration=float(np.median(ts) * 1e3 / max(1, s["iterations"])),
           gpu_residuals_per_s=2.0 * n_obs * s["iterations"] / float(np.median(ts)),
           oracle_iterations=o[5]["iterations"], oracle_solve_s=t_cpu,
           cam_t_max_diff=float(np.abs(g[1] - o[1]).max()), frame_t_max_diff=float(np.abs(g[3] - o[3]).max()),
           final_cost_gpu=s["final_cost"], final_cost_oracle=o[5]["final_cost"])
print(json.dumps(out))
