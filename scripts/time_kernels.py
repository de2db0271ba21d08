"""Isolated timing of the decide + elim + solve kernel of one intrinsics LM iteration (developer aid; with an
ablation build in CC_LIB_PATH: how far into the kernel the time goes, scripts/ablate_decide.sh)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from camera_calibrator_amd import capi

F, M = int(os.environ.get("F", 1000)), int(os.environ.get("M", 500))
off, uv, xyz = capi.make_intrinsics_problem(F, M)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
prob.solve(capi.default_options(max_iterations=2))
out = {}
for name, which in (("decide_elim_solve", 1),):
    ts = []
    for _ in range(3):
        ms = C.c_double()
        rc = capi.lib().cc_intrinsics_profile_kernel(prob._h, C.c_int32(which), C.c_int32(200), C.byref(ms))
        assert rc == 0, capi.lib().cc_last_error()
        ts.append(round(ms.value * 1e3, 2))
    out[name] = ts
print(os.path.basename(os.environ.get("CC_LIB_PATH", "product")), "F", F, "M", M, out)
