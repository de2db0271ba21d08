"""Where a round of the persistent intrinsics kernel goes: wall-clock marks (100 MHz, one counter for the chip) left in
round CC_PERSIST_TIMING_ROUND by a worker workgroup in the middle of the grid (a leader) and by the control workgroup of a timing-only build
(scripts/build_variant.sh ptime cc_intrinsics_persist.hip --patch timing -DCC_PERSIST_TIMING; CC_LIB_PATH=scripts/ablate_build/libcc_ptime.so).
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
rows, skews = [], []
G = None
for _ in range(15):
    prob.reset()
    prob.solve(capi.default_options(max_iterations=6, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0), log_capacity=0)
    buf = np.zeros(32)
    capi._check(capi.lib().cc_intrinsics_debug_fetch(prob._h, b"vec_solve", buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(32)))
    rows.append((buf - buf[0]) / 100.0)
    teams = int(os.environ.get("CC_INTR_PERSIST_TEAMS", 0)) or (1 if F + 1 <= 256 else 2 if (F + 1) // 2 + 1 <= 256 else 4)
    G = (F + teams - 1) // teams
    st = np.zeros(G * 4)
    capi._check(capi.lib().cc_intrinsics_debug_fetch(prob._h, b"stats", st.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(G * 4)))
    skews.append((st.reshape(G, 4) - buf[0]) / 100.0)
prob.close()
t = np.median(np.array(rows), axis=0)
wn = ["round start", "pose step + Plus done", "main loop starts", "main loop done (this wave)", "all waves done", "block reduced, statistics",
      "statistics row stored", "assumed elimination + row (+ leader sum) done", "elimination done", "elimination row stored",
      "broadcast received (step if the assumption held)", "leader: sixteen rows gathered", "after a miss: step received",
      "elimination: 6x6 factor done", "elimination: substitutions done"]
cn = ["waits for statistics rows", "rows gathered", "decision taken", "decision stored / skipped", "leader rows gathered", "solve done", "step stored",
      "solve: gradient maximum", "solve: rows built", "solve: factorisation + substitutions", "solve: tests, log record, flags"]
sk = np.median(np.array(skews), axis=0)     # [G][4]: round start, statistics stored, elimination row stored, broadcast received
names = ["round start", "statistics stored", "elimination row stored", "broadcast received"]
print(json.dumps({"frames": F, "pts": M, "workgroups": G, "per_workgroup_marks_us": {n: {"min": round(float(sk[:, i].min()), 2), "median": round(float(np.median(sk[:, i])), 2),
      "max": round(float(sk[:, i].max()), 2)} for i, n in enumerate(names)}}))
if t[27] > 0:   # (-DCC_PERSIST_PROBE_ALLGATHER: every worker also gathered the leaders' rows itself)
    print(json.dumps({"probe": "all workers gather the leaders' rows", "middle worker's waves 1..3 had every row at (us)": [round(float(x), 2) for x in t[27:30]]}))
print(json.dumps({"frames": F, "pts": M, "round_us": round(float(max(t[10], t[12])), 2),
                  "worker": {n: round(float(t[i]), 2) for i, n in enumerate(wn) if t[i] > -1e6},
                  "control": {n: round(float(t[16 + i]), 2) for i, n in enumerate(cn)}}))
