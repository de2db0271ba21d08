"""Independent pin of the rig oracle's ROBUST minimiser, with the Huber loss active at the fixed point.

The reference's rig test asserts nothing, and in its scenario (noise +-0.004 per axis against
a = 3/500 = 0.006) no residual is in the linear tail of the loss at the minimiser: the corrector of the oracle
(and of the kernels) would then only ever be validated by itself. Here >= 5 % of the residual blocks are planted
outliers that STAY in the tail at the solution, and the oracle's converged point is checked against a restatement
of the objective that shares no code with it:

    cost(x) = 1/2 sum_k rho(|r_k|^2),  rho(s) = s (s <= a^2), 2 a sqrt(s) - a^2 (s > a^2)      [ceres::HuberLoss]
    r_k = ( x0/x2 - u, x1/x2 - v ),  x = R(q_cr) (R(q_rw) X + t_rw) + t_cr
                                              [/root/reference/src/extrinsics_calibrator.cpp:57-80,175-176]

(numpy, quaternions normalised inside the rotation as ceres::QuaternionRotatePoint does). Asserted: the oracle's
reported cost and per-observation costs equal the restatement's at its converged point; the central-difference
gradient through QuaternionManifold::Plus vanishes there; scipy.optimize.minimize started there does not lower the
cost. CPU only."""
import numpy as np
import scipy.optimize

from oracle import pyoracle as po
from tests.helpers import quat_plus, rig_outlier_case

A = float(np.float32(3.0) / np.float32(500.0))   # extrinsics_calibrator.cpp:176 (float literal)
TIGHT = dict(function_tolerance=1e-16, gradient_tolerance=1e-14, parameter_tolerance=1e-15, max_iterations=500)


def _rot(q, X):
    """R(q / |q|) X for arrays q [n, 4] (w x y z), X [n, 3]."""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, v = q[:, :1], q[:, 1:]
    t = 2.0 * np.cross(v, X)
    return X + w * t + np.cross(v, t)


def _block_costs(sc, cam_q, cam_t, frame_q, frame_t):
    """1/2 rho(|r_k|^2) per observation and |r_k|^2, numpy restatement (see the module docstring)."""
    off = sc["frame_offsets"]
    frame_of = np.repeat(np.arange(len(off) - 1), np.diff(off))
    c = sc["obs_cam"].astype(np.int64)
    X = sc["world_xyz"].astype(np.float64)[sc["obs_world"].astype(np.int64)]
    uv = sc["obs_uv"].astype(np.float64)
    Xr = _rot(frame_q[frame_of], X) + frame_t[frame_of]
    x = _rot(cam_q[c], Xr) + cam_t[c]
    r = np.stack([x[:, 0] / x[:, 2] - uv[:, 0], x[:, 1] / x[:, 2] - uv[:, 1]], axis=1)
    s = (r * r).sum(axis=1)
    rho = np.where(s <= A * A, s, 2.0 * A * np.sqrt(s) - A * A)
    return 0.5 * rho, s


class _Tangent:
    """x = Plus(x*, delta): 6 tangent coordinates per optimised camera and per frame around a base point."""

    def __init__(self, sc, cq, ct, fq, ft):
        self.sc, self.cq, self.ct, self.fq, self.ft = sc, cq, ct, fq, ft
        self.free_cams = [c for c in range(len(cq)) if not sc["cam_frozen"][c]]
        self.n = 6 * (len(self.free_cams) + len(fq))

    def point(self, d):
        cq, ct, fq, ft = self.cq.copy(), self.ct.copy(), self.fq.copy(), self.ft.copy()
        k = 0
        for c in self.free_cams:
            cq[c] = quat_plus(self.cq[c], d[k:k + 3]); ct[c] = self.ct[c] + d[k + 3:k + 6]; k += 6
        for f in range(len(fq)):
            fq[f] = quat_plus(self.fq[f], d[k:k + 3]); ft[f] = self.ft[f] + d[k + 3:k + 6]; k += 6
        return cq, ct, fq, ft

    def cost(self, d):
        return float(_block_costs(self.sc, *self.point(d))[0].sum())

    def grad(self, d, h=2e-7):
        """Central differences at h and h/2, Richardson-extrapolated (the third derivatives along the rotations are
        large: at h = 1e-6 the plain central difference is off by 1.5e-6, a hundred times the gradient left at the
        converged point)."""
        g = np.zeros(self.n)
        for i in range(self.n):
            e = np.zeros(self.n); e[i] = h
            g1 = (self.cost(d + e) - self.cost(d - e)) / (2 * h)
            g2 = (self.cost(d + 0.5 * e) - self.cost(d - 0.5 * e)) / h
            g[i] = (4.0 * g2 - g1) / 3.0
        return g


def _solve(sc, **kw):
    return po.rig_solve(len(sc["cam_q0"]), sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"],
                        sc["cam_q0"], sc["cam_t0"], sc["cam_frozen"], sc["frame_q0"], sc["frame_t0"],
                        options=po.default_options(**kw))


def test_rig_oracle_fixed_point_with_huber_active():
    sc = rig_outlier_case(3, 24, 8)
    cq, ct, fq, ft, obs_cost, s = _solve(sc, **TIGHT)
    assert s["termination"] in ("GRADIENT", "PARAMETER", "FUNCTION") and s["final_cost"] < s["initial_cost"]
    half_rho, sq = _block_costs(sc, cq, ct, fq, ft)
    tail = sq > A * A
    # the loss is ACTIVE at the minimiser: more than 5 % of the blocks in the linear tail, and those are the planted ones
    assert tail.mean() > 0.05 and tail[sc["outlier"]].mean() > 0.9
    # 1. the oracle's objective IS the restated one (total and per observation; extrinsics_calibrator.cpp:219-225)
    assert np.isclose(half_rho.sum(), s["final_cost"], rtol=1e-12)
    assert np.allclose(obs_cost, half_rho, rtol=1e-10, atol=1e-18)
    # 2. first-order optimality of the restated objective at the oracle's point, through Plus
    tg = _Tangent(sc, cq, ct, fq, ft)
    g = tg.grad(np.zeros(tg.n))
    g0 = _Tangent(sc, sc["cam_q0"], sc["cam_t0"], sc["frame_q0"], sc["frame_t0"]).grad(np.zeros(tg.n))
    # (the solve stops on the function tolerance, at the resolution of the cost in double precision: |g| ~ 1e-8 is a
    # predicted decrease of g^2 / 2H ~ 1e-18, against an initial gradient of ~2)
    assert np.abs(g).max() < 1e-7 and np.abs(g).max() < 1e-7 * np.abs(g0).max(), (np.abs(g).max(), np.abs(g0).max())
    assert np.abs(g).max() < 10 * max(s["log"][-1]["gradient_max_norm"], 1e-9)   # and it is the gradient the oracle reports
    # 3. a general-purpose minimiser started there finds nothing lower
    res = scipy.optimize.minimize(tg.cost, np.zeros(tg.n), jac=tg.grad, method="L-BFGS-B", options=dict(maxiter=50, ftol=1e-15, gtol=1e-12))
    assert res.fun >= s["final_cost"] * (1 - 1e-9), (res.fun, s["final_cost"])
    # 4. and the L2 minimiser is a different, worse point for this objective (the test can tell the two apart)
    cq2, ct2, fq2, ft2, _, s2 = po.rig_solve(3, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"],
                                              sc["cam_q0"], sc["cam_t0"], sc["cam_frozen"], sc["frame_q0"], sc["frame_t0"],
                                              huber_a=1e6, options=po.default_options(**TIGHT))
    assert _block_costs(sc, cq2, ct2, fq2, ft2)[0].sum() > s["final_cost"] * (1 + 1e-3)
    assert np.abs(ct2 - ct).max() > 1e-5


def test_rig_oracle_default_options_stop_near_that_fixed_point():
    """With the reference's options (function_tolerance 1e-6, extrinsics_calibrator.cpp:206-213) the robust solve
    stops close to the same point: cost within 1e-5 relative of the tightly converged one."""
    sc = rig_outlier_case(3, 24, 8)
    tight = _solve(sc, **TIGHT)
    dflt = _solve(sc, max_iterations=1000)
    assert dflt[5]["final_cost"] >= tight[5]["final_cost"] * (1 - 1e-12)
    assert dflt[5]["final_cost"] <= tight[5]["final_cost"] * (1 + 1e-5)


def test_rig_oracle_reports_the_gradient_norm_ceres_reports():
    """Ceres tests its gradient tolerance on ||x - Plus(x, -g)||_inf (TrustRegionMinimizer), not on the tangent gradient:
    for a quaternion block that is q - Plus(q, -g_rot), four numbers instead of three. The oracle (and the kernels,
    cc_common.hpp pose_grad_proj_max) evaluate that expression in a cancellation-free form while |g_rot| < 1/4; here it is
    formed LITERALLY -- numerical tangent gradient of the restated objective, numpy Plus, subtraction -- at the point the
    oracle stops at after a few iterations, and compared with what the oracle logs for that point."""
    sc = rig_outlier_case(3, 24, 8)
    checked = distinct = 0
    for iters in range(2, 12):
        cq, ct, fq, ft, _, s = _solve(sc, max_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        last = s["log"][-1]
        if not last["accepted"] or not (1e-5 < last["gradient_max_norm"] < 0.1):
            continue
        tg = _Tangent(sc, cq, ct, fq, ft)
        g = tg.grad(np.zeros(tg.n))
        literal, tangent, k = 0.0, np.abs(g).max(), 0
        blocks = [(cq[c], ct[c]) for c in tg.free_cams] + [(fq[f], ft[f]) for f in range(len(fq))]
        for q, _t in blocks:
            literal = max(literal, np.abs(q - quat_plus(q, -g[k:k + 3])).max(), np.abs(g[k + 3:k + 6]).max())
            k += 6
        assert np.isclose(last["gradient_max_norm"], literal, rtol=1e-5, atol=1e-9), (iters, last["gradient_max_norm"], literal, tangent)
        checked += 1
        distinct += abs(literal - tangent) > 2e-5 * tangent   # (a rotation block carries the maximum: the two norms differ)
    assert checked >= 2 and distinct >= 1
