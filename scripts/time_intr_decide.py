"""Where the time of k_intr_decide_elim goes: wall-clock marks (100 MHz) of the block that arrives last -- the critical
path -- left by a timing-only build (scripts/build_variant.sh intrtime cc_intrinsics.hip --patch timing -DCC_INTR_TIMING;
CC_LIB_PATH=scripts/ablate_build/libcc_intrtime.so). Env F, M. Stage durations in microseconds, median over solves
(marks of the last launch of a solve that runs MAXIT iterations without converging checks)."""
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
rows, srows = [], []
for _ in range(9):
    prob.reset()
    # 3 iterations: the solve stops on the iteration limit, so its last launch is a full decide + elim + solve step
    prob.solve(capi.default_options(max_iterations=3, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0), log_capacity=0)
    buf = np.zeros(64)
    capi._check(capi.lib().cc_intrinsics_debug_fetch(prob._h, b"vec_solve", buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(64)))
    rows.append(np.diff(buf[32:41] / 100.0))
    srows.append(np.diff(buf[49:56] / 100.0))   # (mark 0 belongs to the solve's last, empty launch: the stages start at the gather barrier)
prob.close()
d = np.median(np.array(rows), axis=0)
names = ["kernarg + gather + statistics", "decision (thread 0) + barrier", "block loads + 6x6 Cholesky", "substitutions + staging",
         "Schur sums (+ later frame passes)", "partial store + arrival", "row reads + sums", "9x9 solve + tests + publication"]
sd = np.median(np.array(srows), axis=0)
snames = ["pose back-substitution", "Plus + rotation (one lane)", "first pass (wave 0)",
          "remaining passes + barrier", "cross-wave reduction", "block store + statistics"]
print(json.dumps({"kernel": "k_intr_sweep (middle workgroup, from the gather barrier on)", "frames": F, "pts": M, "total_us": float(sd.sum()),
                  **{n: round(float(v), 2) for n, v in zip(snames, sd)}}))
print(json.dumps({"kernel": "k_intr_decide_elim (last block)", "frames": F, "pts": M, "total_us": float(d.sum()),
                  **{n: round(float(v), 2) for n, v in zip(names, d)}}))
