"""Runs a handful of LM solves of the bench workload (no timing) so that rocprofv3 --pmc can count
HBM traffic of the Jacobian sweep. Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 scripts/prof_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (loads the ROCm runtime first, as bench.py does)
from camera_calibrator_amd import capi

off, uv, xyz = capi.make_intrinsics_problem(1000, 500)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
for _ in range(10):
    prob.reset()
    s = prob.solve(capi.default_options(use_graph=0))
print("iterations", s["iterations"], "cost", s["final_cost"])
prob.close()
