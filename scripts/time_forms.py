"""Both forms of the intrinsics solver on one box: microseconds per counted LM iteration of complete solves from the Zhang
initialisation (reference options), for the frame counts of BASELINE configs[2] and its shards. One process per form
(CC_INTR_PERSIST is read when a handle is created). Prints one JSON line per (frames, form)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, json, os
sys.path.insert(0, %r)
import numpy as np, torch
from camera_calibrator_amd import capi
F = int(sys.argv[1])
off, uv, xyz = capi.make_intrinsics_problem(F, 500)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
o = capi.default_options()
for _ in range(300):
    prob.reset(); prob.solve_lean(o)
ts = []
for _ in range(200):
    prob.reset()
    t = time.perf_counter(); its = prob.solve_lean(o); ts.append((time.perf_counter() - t) * 1e6 / its)
print(json.dumps({"frames": F, "pts": 500, "form": prob.solver_form(), "iterations": its, "us_per_iteration_median": float(np.median(ts)),
                  "us_per_iteration_min": float(np.min(ts))}))
prob.close()
''' % ROOT
for F in (1000, 500, 250, 125):
    for persist in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", CODE, str(F)], env=dict(os.environ, CC_INTR_PERSIST=persist), stdout=subprocess.PIPE, text=True)
        sys.stdout.write(r.stdout)
        sys.stdout.flush()
