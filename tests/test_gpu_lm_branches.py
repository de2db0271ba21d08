"""Rarely taken branches of the device-side trust-region state machine, forced on purpose and compared
with the oracle: rejected steps, non-monotonic acceptance off, every termination reason, invalid
(non-finite) inputs. Costs 1e-9 relative, identical accept/reject sequences and termination."""
import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import intrinsics_case

pytestmark = pytest.mark.gpu


def _both(case, intr0=None, q0=None, t0=None, **kw):
    intr0 = case["intr0"] if intr0 is None else intr0
    q0 = case["q0"] if q0 is None else q0
    t0 = case["t0"] if t0 is None else t0
    g = capi.intrinsics_optimize(case["off"], case["uv"], case["xyz"], intr0, q0, t0, options=capi.default_options(**kw))
    o = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], intr0, q0, t0, options=po.default_options(**kw))
    return g, o


def _same_trajectory(g, o, rtol=1e-9):
    sg, so = g[3], o[3]
    assert sg["termination"] == so["termination"], (sg["termination"], so["termination"])
    assert sg["iterations"] == so["iterations"] and sg["successful_steps"] == so["successful_steps"]
    assert [l["accepted"] for l in sg["log"]] == [l["accepted"] for l in so["log"]]
    assert [l["valid"] for l in sg["log"]] == [l["valid"] for l in so["log"]]
    assert np.allclose([l["cost"] for l in sg["log"]], [l["cost"] for l in so["log"]], rtol=rtol)
    assert np.allclose([l["radius"] for l in sg["log"]], [l["radius"] for l in so["log"]], rtol=1e-6)
    assert np.allclose(g[0], o[0], rtol=1e-8, atol=1e-10)


def _bad_start(case, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    intr = case["intr0"].copy()
    intr[:2] *= 1.0 + 0.35 * scale
    intr[4:] = np.array([0.3, -0.2, 0.02, -0.02, 0.1]) * scale
    q = case["q0"] + 0.15 * scale * rng.normal(size=case["q0"].shape)
    t = case["t0"] * (1.0 + 0.25 * scale * rng.normal(size=case["t0"].shape))
    return intr, q, t


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("kw", [
    dict(min_relative_decrease=0.99, initial_radius=1e8),
    dict(min_relative_decrease=0.999, max_consecutive_invalid_steps=3),
    dict(min_relative_decrease=0.99, use_nonmonotonic_steps=0),
])
def test_rejected_steps_and_radius_shrinking(seed, kw):
    # a demanding acceptance threshold rejects most steps of a moderately bad start: the reject
    # branch (radius /= nu, nu *= 2), its reset on acceptance and the non-monotonic bookkeeping all run
    case = intrinsics_case(12, 60)
    intr, q, t = _bad_start(case, seed, 0.3)
    g, o = _both(case, intr, q, t, max_iterations=40, **kw)
    acc = [l["accepted"] for l in o[3]["log"]]
    assert acc.count(0) >= 10 and acc.count(1) >= 5, "the scenario is meant to mix accepted and rejected steps"
    _same_trajectory(g, o, rtol=1e-8)


def test_wild_start_terminates_with_a_comparable_answer():
    # far outside the basin (initial cost ~1e9, points close to the camera plane): trajectories are
    # chaotic there, so only termination and the recovered optimum are compared
    case = intrinsics_case(12, 60)
    intr, q, t = _bad_start(case, 0, 1.0)
    g, o = _both(case, intr, q, t, initial_radius=1e12, max_iterations=200)
    assert g[3]["termination"] == o[3]["termination"] == "FUNCTION"
    assert np.isclose(g[3]["final_cost"], o[3]["final_cost"], rtol=1e-6)


def test_without_jacobi_scaling():
    case = intrinsics_case(12, 60)
    g, o = _both(case, jacobi_scaling=0, max_iterations=30)
    _same_trajectory(g, o)


@pytest.mark.parametrize("kw,term", [
    (dict(gradient_tolerance=1e12), "GRADIENT"),                 # met by the initial point: 0 iterations
    (dict(parameter_tolerance=1e-1), "PARAMETER"),               # first step is "small enough"
    (dict(function_tolerance=0.999), "FUNCTION"),
    (dict(max_iterations=1), "NO_CONVERGENCE"),
    (dict(initial_radius=1e-40, min_radius=1e-32), "MIN_RADIUS"),
])
def test_every_termination_reason(kw, term):
    case = intrinsics_case(8, 50)
    g, o = _both(case, **kw)
    assert o[3]["termination"] == term
    _same_trajectory(g, o)


def test_gradient_tolerance_met_after_some_iterations():
    case = intrinsics_case(8, 50)
    g, o = _both(case, gradient_tolerance=5.0, function_tolerance=-1.0, parameter_tolerance=-1.0, max_iterations=50)
    assert o[3]["termination"] == "GRADIENT" and o[3]["iterations"] >= 1
    _same_trajectory(g, o)


def test_non_finite_observation_fails_the_same_way():
    case = intrinsics_case(6, 40)
    uv = case["uv"].copy()
    uv[17, 0] = np.nan
    bad = dict(case, uv=uv)
    g, o = _both(bad, max_iterations=20)
    assert g[3]["termination"] == o[3]["termination"] and g[3]["iterations"] == o[3]["iterations"]
    assert np.array_equal(np.isnan(g[0]), np.isnan(o[0]))


def test_point_behind_camera_and_huge_radius_do_not_hang():
    case = intrinsics_case(6, 40)
    t = case["t0"].copy()
    t[2, 2] = -t[2, 2]                      # one board behind the camera: a legal (mirrored) configuration
    g, o = _both(case, None, None, t, max_iterations=40, initial_radius=1e16)
    assert g[3]["termination"] == o[3]["termination"]
    assert np.isclose(g[3]["log"][0]["cost"], o[3]["log"][0]["cost"], rtol=1e-6)
    assert np.isfinite(g[3]["final_cost"]) and np.isfinite(g[0]).all()


# ---- the same state machine behind the rig path ---------------------------------------------

def _rig_both(sc, n_cams, **kw):
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    args = (n_cams, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct,
            sc["cam_frozen"], fq, ft)
    g = capi.rig_optimize(*args, options=capi.default_options(**kw))
    o = po.rig_solve(*args, options=po.default_options(**kw))
    return g, o


def _rig_same(g, o):
    sg, so = g[5], o[5]
    assert sg["termination"] == so["termination"], (sg["termination"], so["termination"])
    assert sg["iterations"] == so["iterations"] and sg["successful_steps"] == so["successful_steps"]
    assert [l["accepted"] for l in sg["log"]] == [l["accepted"] for l in so["log"]]
    assert np.allclose([l["cost"] for l in sg["log"]], [l["cost"] for l in so["log"]], rtol=1e-8)
    for k in range(4):
        assert np.abs(g[k] - o[k]).max() < 1e-8


@pytest.mark.parametrize("kw", [
    dict(min_relative_decrease=1.45, use_nonmonotonic_steps=0, max_iterations=40),   # AAA then shrinking rejections
    dict(min_relative_decrease=1.8, max_iterations=40),                              # nothing is ever good enough
])
def test_rig_rejected_steps(kw):
    # Huber-robustified steps of this scenario gain more than the model predicts (rho ~ 1.4-1.8), so
    # the threshold has to sit above 1 to exercise the reject branch and the shrinking radius
    sc = po.rig_scenario(3, 30, 12)
    g, o = _rig_both(sc, 3, **kw)
    acc = [l["accepted"] for l in o[5]["log"]]
    assert acc.count(0) >= 5 and o[5]["termination"] == "PARAMETER"
    _rig_same(g, o)


@pytest.mark.parametrize("kw,term", [
    (dict(gradient_tolerance=1e12), "GRADIENT"),
    (dict(parameter_tolerance=1e-1), "PARAMETER"),
    (dict(function_tolerance=0.999), "FUNCTION"),
    (dict(max_iterations=2), "NO_CONVERGENCE"),
    (dict(initial_radius=1e-40, min_radius=1e-32), "MIN_RADIUS"),
])
def test_rig_every_termination_reason(kw, term):
    sc = po.rig_scenario(2, 30, 8)
    g, o = _rig_both(sc, 2, **kw)
    assert o[5]["termination"] == term
    _rig_same(g, o)
