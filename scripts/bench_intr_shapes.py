"""The intrinsics solve at the BASELINE.json shapes that bench.py does not time (configs[0] 20 x 88, configs[1] 200 x 200) next to
configs[2] (1000 x 500): complete solves from the Zhang start on a resident handle, wall time per solve and per LM iteration."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi

for F, M in ((20, 88), (200, 200), (1000, 500)):
    off, uv, xyz = capi.make_intrinsics_problem(F, M)
    K0, q0, t0 = capi.zhang_init(off, uv, xyz)
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    prob = capi.IntrinsicsProblem(off, uv, xyz)
    prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
    s = prob.solve(log_capacity=0)
    ts = []
    for _ in range(200):
        prob.reset()
        a = time.perf_counter()
        s = prob.solve(log_capacity=0)
        ts.append(time.perf_counter() - a)
    prob.close()
    med = float(np.median(ts))
    print(json.dumps(dict(frames=F, pts=M, observations=F * M, iterations=s["iterations"], termination=s["termination"], solve_us=round(med * 1e6, 1),
                          us_per_iteration=round(med * 1e6 / max(1, s["iterations"]), 2), residuals_per_s=2.0 * F * M * s["iterations"] / med)))
