"""EXPERIMENT (not the product path): the first LM step of the single-camera intrinsics problem obtained by a DENSE
Cholesky factorisation of the damped (6F+9) x (6F+9) normal equations on the f64 matrix cores
(scripts/dense_mfma/dense_step.hip), next to the product's exact block-Schur step.

What is compared (same Gram blocks, produced by the product's own sweep `cc_intrinsics_eval`):
  1. dense step (GPU, this experiment)  vs  block-Schur step restated in numpy float64 on the host;
  2. the state the dense step leads to   vs  the state after ONE iteration of the product's solver
     (`cc_intrinsics_solve`, max_iterations = 1), i.e. against the product's device Schur step itself.
What is timed: the dense factorisation + substitutions (hipEvents, average of REPS) and one LM iteration of the product.

Usage (GPU box, repo root):  bash scripts/dense_mfma/build.sh && python scripts/dense_mfma/run_dense.py [F] [M]
Prints one JSON line. F = 1000, M = 500 is BASELINE.json configs[2] (n = 6009)."""
import ctypes as C
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime the product library links against)
from camera_calibrator_amd import capi
from camera_calibrator_amd.harness import quat_plus

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 500
REPS = int(os.environ.get("REPS", 3))
FP64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: dense f64 matrix rate (= the f64 vector rate on gfx950)

off, uv, xyz = capi.make_intrinsics_problem(F, M)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
q0, t0 = q0.astype(np.float64), t0.astype(np.float64)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0, t0)
opt = capi.default_options(max_iterations=1)


def time_product():
    """us per LM iteration over complete solves from the Zhang start, as bench.py measures them"""
    o4 = capi.default_options()
    for _ in range(20):
        prob.reset(); prob.solve(o4, log_capacity=0)
    t_a = time.perf_counter(); its = 0
    for _ in range(100):
        prob.reset(); its += prob.solve(o4, log_capacity=0)["iterations"]
    return (time.perf_counter() - t_a) / its * 1e6


us_product = time_product()   # timed BEFORE the host-side numpy work below: its BLAS threads keep spinning on the box's
prob.reset()                  # CPU quota for a while and slow down the host thread that drives the solver
cost0, G = prob.eval()                                   # [F][16][16], rows/cols = [intr(9) pose(6) r]

# ---- damped, Jacobi-scaled normal equations exactly as the solver forms them (first iteration: radius = initial) ----
radius, dmin, dmax = opt.initial_radius, opt.min_lm_diagonal, opt.max_lm_diagonal
Hss = G[:, :9, :9].sum(0); gs = G[:, :9, 15].sum(0)
Hpp = G[:, 9:15, 9:15].copy(); Hps = G[:, 9:15, :9].copy(); gp = G[:, 9:15, 15].copy()
ss = 1.0 / (1.0 + np.sqrt(np.diag(Hss)))
sp = 1.0 / (1.0 + np.sqrt(np.einsum("fii->fi", Hpp)))
App = sp[:, :, None] * Hpp * sp[:, None, :]
Aps = sp[:, :, None] * Hps * ss[None, None, :]
Ass = ss[:, None] * Hss * ss[None, :]
bp, bs = sp * gp, ss * gs
idx = np.arange(6)
App[:, idx, idx] += np.clip(App[:, idx, idx], dmin, dmax) / radius
Ass[np.arange(9), np.arange(9)] += np.clip(np.diag(Ass), dmin, dmax) / radius

# ---- (a) block-Schur step on the host, float64 ----
Y = np.linalg.solve(App, np.concatenate([Aps, bp[:, :, None]], axis=2))     # [F][6][10]
Sred = Ass - np.einsum("fij,fik->jk", Aps, Y[:, :, :9])
bred = bs - np.einsum("fij,fi->j", Aps, Y[:, :, 9])
ds_schur = -np.linalg.solve(Sred, bred)
dp_schur = -(Y[:, :, 9] + Y[:, :, :9] @ ds_schur)

# ---- (b) the same system as ONE dense matrix: poses first, shared block last ----
n = 6 * F + 9
A = np.zeros((n, n))
for f in range(F):
    A[6 * f:6 * f + 6, 6 * f:6 * f + 6] = App[f]
    A[6 * F:, 6 * f:6 * f + 6] = Aps[f].T
    A[6 * f:6 * f + 6, 6 * F:] = Aps[f]
A[6 * F:, 6 * F:] = Ass
b = -np.concatenate([bp.reshape(-1), bs])
lib = C.CDLL(os.path.join(HERE, "libdense_step.so"))
d = np.zeros(n)
ms_f, ms_b, flop = C.c_double(), C.c_double(), C.c_double()
pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rc = lib.dn_solve(C.c_int(0), C.c_int(n), pd(A), pd(b), pd(d), C.c_int(REPS), C.byref(ms_f), C.byref(ms_b), C.byref(flop))
assert rc == 0, f"dn_solve failed ({rc})"
dp_dense, ds_dense = d[:6 * F].reshape(F, 6), d[6 * F:]
scale = max(np.abs(ds_schur).max(), np.abs(dp_schur).max())
err_host = max(np.abs(ds_dense - ds_schur).max(), np.abs(dp_dense - dp_schur).max()) / scale
resid = np.abs(A @ d - b).max() / np.abs(b).max()

# ---- (c) against the product: one iteration of the device solver from the same state ----
t1 = time.perf_counter()
s = prob.solve(opt, log_capacity=4)
intr1, q1, t1s = prob.get_state()
# candidate state the dense step leads to (unscale, then Plus)
intr_d = intr0 + ss * ds_dense
dpu = sp * dp_dense
q_d = np.stack([quat_plus(q0[f], dpu[f, :3]) for f in range(F)])
t_d = t0 + dpu[:, 3:]
accepted = bool(s["log"][-1]["accepted"]) if s["log"] else None
err_prod = max(np.abs(intr_d - intr1).max() / np.abs(intr1).max(), np.abs(q_d - q1).max(), np.abs(t_d - t1s).max() / np.abs(t1s).max())
prob.close()

tflops = flop.value / (ms_f.value * 1e-3) / 1e12
print(json.dumps({
    "experiment": "dense (6F+9)^2 Cholesky of the damped normal equations on v_mfma_f64_16x16x4_f64 vs the product's block-Schur step",
    "frames": F, "points_per_frame": M, "n": n, "matrix_MB": n * n * 8 / 1e6,
    "dense_factor_ms": ms_f.value, "dense_backsub_ms": ms_b.value, "dense_total_ms": ms_f.value + ms_b.value,
    "mfma_flop_per_factorisation": flop.value, "useful_flop_n3_over_3": n ** 3 / 3.0,
    "mfma_tflops": tflops, "mfma_frac_of_f64_peak": tflops / FP64_PEAK_TFLOPS,
    "product_us_per_lm_iteration": us_product,
    "dense_over_product": (ms_f.value + ms_b.value) * 1e3 / us_product,
    "step_rel_err_vs_host_schur": err_host, "dense_residual_rel": resid,
    "state_rel_err_vs_product_iteration": err_prod, "product_step_accepted": accepted,
}))
