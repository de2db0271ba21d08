"""Bits of one solve with the library CC_LIB_PATH names (or the in-tree one): cost history + final state of the rig scenario
C x F x M, saved as .npz -- two libraries that should compute the same thing are compared file against file.
    C=8 F=2000 M=64 OUT=gpurun_out/x/a.npz python scripts/ab_bits.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi

C, F, M = int(os.environ.get("C", 8)), int(os.environ.get("F", 2000)), int(os.environ.get("M", 64))
sc = capi.rig_scenario(C, F, M)
cq, ct = capi.affine_to_qt(sc["cam_T"])
fq, ft = capi.affine_to_qt(sc["frame_T"])
prob = capi.RigProblem(C, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
prob.set_state(cq, ct, fq, ft)
o = capi.default_options(max_iterations=int(os.environ.get("ITERS", 1000)))
s = prob.solve(o, log_capacity=2048)
st = prob.get_state()
cost = np.array([it["cost"] for it in s["log"]]) if "log" in s else np.zeros(0)
np.savez(os.environ["OUT"], cost=cost, iterations=s["iterations"], final_cost=s["final_cost"], **{"s%d" % i: np.asarray(a) for i, a in enumerate(st)})
print(s["iterations"], s["final_cost"], len(cost))
