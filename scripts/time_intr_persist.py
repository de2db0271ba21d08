"""Where a round of the persistent intrinsics kernel goes: wall-clock marks (100 MHz, one counter for the chip) left in
round CC_PERSIST_TIMING_ROUND by a worker workgroup in the middle of the grid (a leader) and by the control workgroup of a timing-only build
(scripts/build_variant.sh ptime cc_intrinsics_persist.hip -DCC_PERSIST_TIMING; CC_LIB_PATH=scripts/ablate_build/libcc_ptime.so).
Env F, M. Microseconds, median over solves; `t` = time since the worker's round start."""
import ctypes as C, json, os, sys
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
rows = []
for _ in range(15):
    prob.reset()
    prob.solve(capi.default_options(max_iterations=6, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0), log_capacity=0)
    buf = np.zeros(32)
    capi._check(capi.lib().cc_intrinsics_debug_fetch(prob._h, b"vec_solve", buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(32)))
    rows.append((buf - buf[0]) / 100.0)
prob.close()
t = np.median(np.array(rows), axis=0)
wn = ["round start", "pose step + Plus done", "main loop starts", "main loop done (this wave)", "all waves done", "block reduced, statistics",
      "statistics row stored", "decision received", "elimination done", "elimination row stored", "step received",
      "leader: sixteen rows gathered", "leader: row posted"]
cn = ["waits for statistics rows", "rows gathered", "decision taken", "decision stored", "elimination rows gathered", "solve done", "step stored"]
print(json.dumps({"frames": F, "pts": M, "round_us": round(float(t[10]), 2),
                  "worker": {n: round(float(t[i]), 2) for i, n in enumerate(wn)},
                  "control": {n: round(float(t[16 + i]), 2) for i, n in enumerate(cn)}}))
