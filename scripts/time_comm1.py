import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
off, uv, xyz = capi.make_intrinsics_problem(1000, 500)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
def bench(tag, opts):
    prob.reset(); prob.solve(opts, log_capacity=0)
    ts = []
    for _ in range(30):
        prob.reset(); t = time.perf_counter(); s = prob.solve(opts, log_capacity=0); ts.append(time.perf_counter() - t)
    print(tag, "solve ms", round(float(np.median(ts)) * 1e3, 4), "iters", s["iterations"], "us/iter", round(float(np.median(ts)) * 1e6 / s["iterations"], 1))
bench("graph       ", capi.default_options())
bench("no graph    ", capi.default_options(use_graph=0))
prob.exchange_attach(0, [prob.exchange_export()])
bench("mailbox 1 rk", capi.default_options())
prob.close()
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
prob.comm_init(capi.comm_get_unique_id(), 0, 1)
bench("rccl 1 rank ", capi.default_options())
s = prob.solve(capi.default_options(profile_kernels=1), log_capacity=0)
prob.reset(); s = prob.solve(capi.default_options(profile_kernels=1), log_capacity=0)
print({k: (round(s["kernel_ms"][k] / max(1, s["kernel_launches"][k]) * 1e3, 2), s["kernel_launches"][k]) for k in s["kernel_ms"]})
