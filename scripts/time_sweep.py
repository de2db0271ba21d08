import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
F, M = int(os.environ.get("F", 1000)), int(os.environ.get("M", 500))
off, uv, xyz = capi.make_intrinsics_problem(F, M)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
s = prob.solve(capi.default_options(max_iterations=2))
ts = [prob.profile_sweep(200) for _ in range(3)]
print(os.environ.get("CC_LIB_PATH", "product"), "F", F, "M", M, "sweep us", [round(t * 1e3, 2) for t in ts])
