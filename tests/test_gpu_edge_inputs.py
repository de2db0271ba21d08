"""Edge inputs through the C ABI: empty frames, a single frame, no observation at all, malformed
arguments (error code + cc_last_error, never a device fault), use-before-set_state."""
import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import intrinsics_case

pytestmark = pytest.mark.gpu


def _with_empty_frames(case, empties):
    """Insert zero-observation frames (arbitrary poses) at the given positions."""
    off = list(case["off"])
    q, t = list(case["q0"]), list(case["t0"])
    for e in sorted(empties):
        off.insert(e + 1, off[e])
        q.insert(e, np.array([1.0, 0, 0, 0]))
        t.insert(e, np.array([0.0, 0, 1.0]))
    return dict(case, off=np.array(off, dtype=np.int64), q0=np.array(q), t0=np.array(t))


def test_frames_without_observations_are_carried_along_untouched():
    case = _with_empty_frames(intrinsics_case(6, 40), [0, 3, 7])
    g = capi.intrinsics_optimize(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    o = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    assert g[3]["termination"] == o[3]["termination"] and g[3]["iterations"] == o[3]["iterations"]
    assert np.allclose([l["cost"] for l in g[3]["log"]], [l["cost"] for l in o[3]["log"]], rtol=1e-9)
    assert np.allclose(g[0], o[0], rtol=1e-9, atol=1e-10)
    for e in (0, 3, 7):
        assert np.array_equal(g[1][e], case["q0"][e]) and np.array_equal(g[2][e], case["t0"][e])
    assert np.allclose(g[1], o[1], atol=1e-9) and np.allclose(g[2], o[2], atol=1e-9)


def test_single_frame_problem():
    full = intrinsics_case(5, 120)          # K and the pose come from a five-view estimate
    n = int(full["off"][1])
    case = dict(off=full["off"][:2].copy(), uv=full["uv"][:n], xyz=full["xyz"][:n], intr0=full["intr0"],
                q0=full["q0"][:1], t0=full["t0"][:1])
    mask = 0b111110000                      # one view cannot pin the distortion: optimise fx fy px py + pose
    g = capi.intrinsics_optimize(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"], const_mask=mask)
    o = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"], const_mask=mask)
    assert o[3]["iterations"] >= 2 and np.isfinite(o[3]["final_cost"])
    assert g[3]["termination"] == o[3]["termination"] and g[3]["iterations"] == o[3]["iterations"]
    assert np.allclose([l["cost"] for l in g[3]["log"]], [l["cost"] for l in o[3]["log"]], rtol=1e-7)
    assert np.array_equal(g[0][4:], case["intr0"][4:])


def test_problem_without_any_observation():
    off = np.zeros(4, dtype=np.int64)
    uv = np.zeros((0, 2), np.float32)
    xyz = np.zeros((0, 3), np.float32)
    intr = np.array([1000.0, 1000, 500, 500, 0, 0, 0, 0, 0])
    q = np.tile([1.0, 0, 0, 0], (3, 1))
    t = np.tile([0.0, 0, 1], (3, 1))
    g = capi.intrinsics_optimize(off, uv, xyz, intr, q, t)
    o = po.intrinsics_solve(off, uv, xyz, intr, q, t)
    assert g[3]["termination"] == o[3]["termination"] and g[3]["iterations"] == o[3]["iterations"] == 0
    assert g[3]["final_cost"] == 0.0 and np.array_equal(g[0], intr)


def test_malformed_arguments_are_refused_with_a_message():
    case = intrinsics_case(3, 10)
    bad_off = case["off"].copy()
    bad_off[1], bad_off[2] = bad_off[2], bad_off[1]
    with pytest.raises(capi.CcError, match="non-decreasing"):
        capi.IntrinsicsProblem(bad_off, case["uv"], case["xyz"])
    shifted = case["off"] + 1
    with pytest.raises(capi.CcError, match=r"frame_offsets\[0\]"):
        capi.IntrinsicsProblem(shifted, np.vstack([case["uv"], case["uv"][:1]]), np.vstack([case["xyz"], case["xyz"][:1]]))
    with pytest.raises(capi.CcError):
        capi.IntrinsicsProblem(case["off"], case["uv"], case["xyz"], device=99)
    p = capi.IntrinsicsProblem(case["off"], case["uv"], case["xyz"])
    with pytest.raises(capi.CcError, match="no state"):
        p.solve()
    p.close()


def test_rig_malformed_arguments_are_refused():
    sc = po.rig_scenario(2, 5, 4)
    cam_bad = sc["obs_cam"].copy()
    cam_bad[3] = 7                           # camera index out of range
    with pytest.raises(capi.CcError):
        capi.RigProblem(2, sc["frame_offsets"], cam_bad, sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    w_bad = sc["obs_world"].copy()
    w_bad[0] = len(sc["world_xyz"])          # world point index out of range
    with pytest.raises(capi.CcError):
        capi.RigProblem(2, sc["frame_offsets"], sc["obs_cam"], w_bad, sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
    # any number of cameras is fine as long as few enough are optimised (only observed, non-frozen ones own columns)
    capi.RigProblem(11, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"],
                    np.zeros(11, np.uint8)).close()
