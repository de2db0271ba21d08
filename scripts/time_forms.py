"""Both forms of the intrinsics solver on one box: microseconds per counted LM iteration of complete solves from the Zhang
initialisation (reference options), for the frame counts of BASELINE configs[2] and its shards (env PTS: points per frame, several allowed). One process per form
(CC_INTR_PERSIST is read when a handle is created). Prints one JSON line per (frames, form)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, json, os
sys.path.insert(0, %r)
import numpy as np, torch
from camera_calibrator_amd import capi
F, M = int(sys.argv[1]), int(sys.argv[2])
off, uv, xyz = capi.make_intrinsics_problem(F, M)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0.astype(np.float64), t0.astype(np.float64))
o = capi.default_options()
for _ in range(100):
    prob.reset(); prob.solve_lean(o)
ts = []
for _ in range(100):
    prob.reset()
    t = time.perf_counter(); its = prob.solve_lean(o); ts.append((time.perf_counter() - t) * 1e6 / its)
print(json.dumps({"frames": F, "pts": M, "form": prob.solver_form(), "iterations": its, "us_per_iteration_median": float(np.median(ts)),
                  "us_per_iteration_min": float(np.min(ts))}))
prob.close()
''' % ROOT
# (round 6: points per frame as a second axis -- PTS="500 100 20"; the default is the round-3 table, 500 points)
for F in (1000, 500, 250, 125):
  for M in [int(x) for x in os.environ.get("PTS", "500").split()]:
    for persist in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", CODE, str(F), str(M)], env=dict(os.environ, CC_INTR_PERSIST=persist), stdout=subprocess.PIPE, text=True)
        sys.stdout.write(r.stdout)
        sys.stdout.flush()
