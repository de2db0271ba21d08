"""A/B timing of complete C3 solves on one GPU within ONE process run (boxes differ by several per cent):
per-iteration time over individually timed solves. Usage: python scripts/ab_solve.py [label]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi

F, M = int(os.environ.get("F", 1000)), int(os.environ.get("M", 500))
off, uv, xyz = capi.make_intrinsics_problem(F, M)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
o = capi.default_options()
for _ in range(50):
    prob.reset(); s = prob.solve(o, log_capacity=0)
ts = []
for _ in range(300):
    prob.reset()
    t1 = time.perf_counter(); s = prob.solve(o, log_capacity=0); ts.append(time.perf_counter() - t1)
t1 = time.perf_counter()
n = 0
for _ in range(300):
    prob.reset(); s = prob.solve(o, log_capacity=0); n += s["iterations"]
loop = (time.perf_counter() - t1) / n
sweep = prob.profile_sweep(200)
prob.close()
ts = np.array(ts) * 1e6 / s["iterations"]
print(json.dumps(dict(label=sys.argv[1] if len(sys.argv) > 1 else "", us_per_iter_median=float(np.median(ts)), p10=float(np.percentile(ts, 10)),
                      loop_us_per_iter=loop * 1e6, sweep_us=sweep * 1e3, iterations=s["iterations"])))
