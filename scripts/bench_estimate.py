"""Calibrator::Estimate end to end (device Zhang init + device LM) vs the CPU oracle pipeline, C3 size."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camera_calibrator_amd import capi
from oracle import pyoracle as po
F, M = int(os.environ.get("F", 1000)), int(os.environ.get("M", 500))
off, uv, xyz = po.make_intrinsics_problem(F, M)
capi.zhang_init(off, uv, xyz)  # warm
ts = []
for _ in range(5):
    t0 = time.perf_counter(); Kg, qg, tg = capi.zhang_init(off, uv, xyz); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); Ko, qo, to = po.zhang_init(off, uv, xyz); t_cpu_init = time.perf_counter() - t0
intr0 = np.array([Kg[0, 0], Kg[1, 1], Kg[0, 2], Kg[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
capi.intrinsics_optimize(off, uv, xyz, intr0, qg.astype(np.float64), tg.astype(np.float64))  # warm
t0 = time.perf_counter()
ig, _, _, sg = capi.intrinsics_optimize(off, uv, xyz, intr0, qg.astype(np.float64), tg.astype(np.float64))
t_gpu_opt = time.perf_counter() - t0
t0 = time.perf_counter()
io, _, _, so = po.intrinsics_solve(off, uv, xyz, intr0, qo.astype(np.float64), to.astype(np.float64))
t_cpu_opt = time.perf_counter() - t0
print(json.dumps(dict(frames=F, pts=M, gpu_zhang_init_ms=float(np.median(ts) * 1e3), cpu_zhang_init_ms=t_cpu_init * 1e3,
                      gpu_optimize_one_shot_ms=t_gpu_opt * 1e3, cpu_optimize_ms=t_cpu_opt * 1e3,
                      K_rel_diff=float(np.abs(Kg - Ko).max() / 1000), note="one-shot calls include allocation, H2D upload and read-back (steady state, second call)")))
