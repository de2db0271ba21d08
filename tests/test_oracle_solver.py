"""The oracle's LM driver: cross-check of the minimiser against scipy.optimize.least_squares and
size-independent properties."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from oracle import pyoracle as po
from tests.helpers import TIGHT, intrinsics_case, quat_plus


def _scipy_minimise(case, free_mask):
    off, uv, xyz = case["off"], case["uv"].astype(np.float64), case["xyz"].astype(np.float64)
    F = len(off) - 1
    intr0, q0, t0 = case["intr0"], case["q0"], case["t0"]
    free = np.flatnonzero(free_mask)

    def unpack(p):
        intr = intr0.copy(); intr[free] = p[:len(free)]
        d = p[len(free):].reshape(F, 6)
        return intr, d

    def fun(p):
        intr, d = unpack(p)
        out = np.empty(2 * off[-1])
        for f in range(F):
            q = quat_plus(q0[f], d[f, :3])  # local parameterisation around the initial pose
            w, x, y, z = q / np.linalg.norm(q)
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                          [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                          [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            Xc = xyz[off[f]:off[f + 1]] @ R.T + t0[f] + d[f, 3:]
            xn, yn = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
            r2 = xn * xn + yn * yn
            fx, fy, px, py, k1, k2, p1, p2, k3 = intr
            m = 1 + k1 * r2 + k2 * r2**2 + k3 * r2**3
            xd = xn * m + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
            yd = yn * m + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
            sl = slice(2 * off[f], 2 * off[f + 1])
            out[sl] = np.stack([fx * xd + px - uv[off[f]:off[f + 1], 0], fy * yd + py - uv[off[f]:off[f + 1], 1]], 1).ravel()
        return out

    p0 = np.concatenate([intr0[free], np.zeros(6 * F)])
    sol = least_squares(fun, p0, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=200)
    intr, _ = unpack(sol.x)
    return intr, 0.5 * float(np.sum(sol.fun**2))


@pytest.mark.parametrize("mask", [0, 1 << 8])
def test_minimiser_matches_scipy(mask):
    case = intrinsics_case(6, 40)
    io, qo, to, so = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"],
                                         const_mask=mask, options=po.default_options(**TIGHT))
    free = np.array([not (mask >> i) & 1 for i in range(9)])
    isci, cost_sci = _scipy_minimise(case, free)
    assert abs(so["final_cost"] - cost_sci) <= 1e-9 * cost_sci
    assert np.all(np.abs(io[:4] - isci[:4]) <= 1e-6 * np.abs(isci[:4]))
    assert np.all(np.abs(io[4:] - isci[4:]) <= 1e-5 * np.maximum(np.abs(isci[4:]), 1e-3))
    if mask:
        assert io[8] == case["intr0"][8]


def test_zero_noise_recovers_generator_truth():
    case = intrinsics_case(8, 60, noise=0.0)
    io, _, _, so = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"],
                                       options=po.default_options(**TIGHT))
    truth = np.array([1000, 1000, 800, 500, -4.0e-2, 5e-4, 1.0e-3, 2.0e-5, -3e-4])
    # inputs are float32 pixels/points: residual floor ~1e-5 px
    assert so["final_cost"] < 1e-6
    assert np.all(np.abs(io[:4] - truth[:4]) < 2e-2)
    assert np.all(np.abs(io[[4, 6, 7]] - truth[[4, 6, 7]]) < 1e-4)


def test_trajectory_is_deterministic_and_thread_count_independent():
    case = intrinsics_case(20, 88)
    a = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    b = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    assert np.array_equal(a[0], b[0]) and a[3]["final_cost"] == b[3]["final_cost"]
    c = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"],
                            options=po.default_options(num_threads=4))
    assert c[3]["iterations"] == a[3]["iterations"]
    assert np.allclose(c[0], a[0], rtol=1e-10, atol=1e-12)


def test_quaternion_sign_gauge():
    case = intrinsics_case(6, 40)
    a = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"], options=po.default_options(**TIGHT))
    b = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], -case["q0"], case["t0"], options=po.default_options(**TIGHT))
    assert np.allclose(a[0], b[0], rtol=1e-10, atol=1e-12)


def test_default_options_match_reference_call_site():
    o = po.default_options()
    # calibrator.cpp:314-321 + Ceres defaults
    assert (o.max_iterations, o.use_nonmonotonic_steps, o.num_threads) == (100, 1, 1)
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert (o.initial_radius, o.max_radius, o.min_radius, o.min_relative_decrease) == (1e4, 1e16, 1e-32, 1e-3)


def test_log_is_consistent():
    case = intrinsics_case(20, 88)
    _, _, _, s = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    assert s["iterations"] == len(s["log"]) and s["successful_steps"] == sum(l["accepted"] for l in s["log"])
    assert s["final_cost"] <= s["initial_cost"]
    for l in s["log"]:
        if l["accepted"]:
            assert l["relative_decrease"] > 1e-3 and l["model_cost_change"] > 0


# ---- rig ------------------------------------------------------------------------------------

def _rig_inputs(sc):
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    return cq, ct, fq, ft


def test_rig_solve_reduces_cost_and_moves_towards_truth():
    sc = po.rig_scenario(2, 100, 4)     # test_extrinsics_calibrator.cpp scenario, downsized
    cq, ct, fq, ft = _rig_inputs(sc)
    cq1, ct1, fq1, ft1, cost, s = po.rig_solve(2, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"],
                                               sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
    assert s["final_cost"] < 0.05 * s["initial_cost"]
    assert np.array_equal(cq1[0], cq[0]) and np.array_equal(ct1[0], ct[0])      # frozen camera untouched
    tq, tt = po.affine_to_qt(sc["cam_T_true"])
    assert np.abs(ct1[1] - tt[1]).max() < np.abs(ct[1] - tt[1]).max()
    assert cost.shape == (len(sc["obs_cam"]),) and np.isclose(cost.sum(), s["final_cost"], rtol=1e-12)


def test_huber_equals_l2_when_all_residuals_are_small():
    sc = po.rig_scenario(2, 30, 4)
    cq, ct, fq, ft = _rig_inputs(sc)
    args = (2, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct,
            sc["cam_frozen"], fq, ft)
    a = po.rig_solve(*args, huber_a=1e3)
    b = po.rig_solve(*args, huber_a=1e6)
    assert np.allclose(a[1], b[1], rtol=0, atol=1e-13) and a[5]["final_cost"] == b[5]["final_cost"]
    c = po.rig_solve(*args)  # reference constant 3.0f/500.0f: outliers get down-weighted
    assert c[5]["initial_cost"] < a[5]["initial_cost"]


def test_rig_unobserved_camera_and_empty_frame_are_left_alone():
    sc = po.rig_scenario(3, 20, 4)
    keep = sc["obs_cam"] != 2                      # camera 2 never observes anything
    obs_cam, obs_world, obs_uv = sc["obs_cam"][keep], sc["obs_world"][keep], sc["obs_uv"][keep]
    off = np.concatenate([[0], np.cumsum([np.count_nonzero(keep[sc["frame_offsets"][f]:sc["frame_offsets"][f + 1]]) for f in range(20)])])
    # make frame 7 empty
    lo, hi = off[7], off[8]
    sel = np.ones(len(obs_cam), bool); sel[lo:hi] = False
    off2 = off.copy(); off2[8:] -= hi - lo
    cq, ct, fq, ft = _rig_inputs(sc)
    r = po.rig_solve(3, off2, obs_cam[sel], obs_world[sel], obs_uv[sel], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
    assert np.array_equal(r[0][2], cq[2]) and np.array_equal(r[1][2], ct[2])
    assert np.array_equal(r[2][7], fq[7]) and np.array_equal(r[3][7], ft[7])
    assert r[5]["final_cost"] < r[5]["initial_cost"]
