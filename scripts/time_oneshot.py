"""Where the one-shot call (create + set_state + solve + get_state + destroy) spends its time (C3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: F401
from camera_calibrator_amd import capi
off, uv, xyz = capi.make_intrinsics_problem(1000, 500)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
q0, t0 = q0.astype(np.float64), t0.astype(np.float64)
for rep in range(3):
    t = [time.perf_counter()]
    p = capi.IntrinsicsProblem(off, uv, xyz); t.append(time.perf_counter())
    p.set_state(intr0, q0, t0); t.append(time.perf_counter())
    s = p.solve(capi.default_options(use_graph=int(os.environ.get("GRAPH", 1))), log_capacity=0); t.append(time.perf_counter())
    p.get_state(); t.append(time.perf_counter())
    p.close(); t.append(time.perf_counter())
    print("create %.3f set_state %.3f solve %.3f get_state %.3f destroy %.3f ms" % tuple((b - a) * 1e3 for a, b in zip(t, t[1:])))
