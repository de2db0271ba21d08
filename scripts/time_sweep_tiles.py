"""Sweep time of a strong-scaling shard of BASELINE.json configs[2] on ONE GPU, with 1, 2, 4 tiles per frame:
F frames x M points (default 125 x 500 = the share of one of eight GPUs), whole-iteration time alongside."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from camera_calibrator_amd import capi

M = int(os.environ.get("M", 500))
for F in [int(x) for x in os.environ.get("F", "125,250,500,1000").split(",")]:
    off, uv, xyz = capi.make_intrinsics_problem(F, M)
    K0, q0, t0 = capi.zhang_init(off, uv, xyz)
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    for T in os.environ.get("TILES", "1,2,4").split(","):
        os.environ["CC_SWEEP_TILES"] = T
        prob = capi.IntrinsicsProblem(off, uv, xyz)
        prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
        s = prob.solve(log_capacity=0)
        ts = []
        for _ in range(20):
            prob.reset()
            t1 = time.perf_counter()
            s = prob.solve(log_capacity=0)
            ts.append(time.perf_counter() - t1)
        sweep_ms = prob.profile_sweep(200)
        prob.close()
        print(json.dumps(dict(frames=F, pts=M, tiles=int(T), sweep_us=sweep_ms * 1e3, iterations=s["iterations"],
                              us_per_iteration=float(np.median(ts)) * 1e6 / s["iterations"], final_cost=s["final_cost"])), flush=True)
    del os.environ["CC_SWEEP_TILES"]
