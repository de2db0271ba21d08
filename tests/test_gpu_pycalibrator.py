"""The class surface end to end on the GPU: pycalibrator.Calibrator / ExtrinsicsCalibrator
against the oracle pipeline, and the reference's own integration tests re-run on it."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "camera_calibrator_amd"))
pytestmark = pytest.mark.gpu

from oracle import pyoracle as po  # noqa: E402


def _frames(off, uv, xyz):
    return [uv[off[f]:off[f + 1]] for f in range(len(off) - 1)], [xyz[off[f]:off[f + 1]] for f in range(len(off) - 1)]


def test_calibration_estimation_works():
    """src/test_calibrator.cpp:45-75 on the class surface (5 frames x 100 planar points)."""
    import pycalibrator as pc
    off, uv, xyz = po.make_intrinsics_problem(5, 100)
    img, world = _frames(off, uv, xyz)
    c = pc.Calibrator(1600, 1000)
    c.Estimate(img, world)
    K, new_K = po.FIXTURE_K, c.GetK()
    for i in range(3):
        for j in range(3):
            if K[i, j] != 0:
                assert (new_K[i, j] - K[i, j]) / K[i, j] < 0.01 and abs(new_K[i, j] - K[i, j]) / K[i, j] < 0.01
            else:
                assert new_K[i, j] == 0
    assert c.LastStatus() == 0 and c.LastIterations() > 0
    assert c.LastSolverReruns() == 0 and c.LastSolverNote() == ""   # (the solve ran in its usual form: nothing to report)
    # same pipeline through the oracle: float32 outputs equal within 1 ulp... of the LM minimiser;
    # the two Zhang initialisations differ in the last float bits, so compare at 1e-5 relative
    K0, q0, t0 = po.zhang_init(off, uv, xyz)
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    io, _, _, _ = po.intrinsics_solve(off, uv, xyz, intr0, q0.astype(np.float64), t0.astype(np.float64))
    assert np.allclose([new_K[0, 0], new_K[1, 1], new_K[0, 2], new_K[1, 2]], io[:4], rtol=1e-5)
    assert np.allclose(c.GetDistortion(), io[4:], rtol=1e-3, atol=1e-6)


def test_optimize_matches_oracle_bitwise_in_float32():
    import pycalibrator as pc
    off, uv, xyz = po.make_intrinsics_problem(20, 88)
    img, world = _frames(off, uv, xyz)
    K0, q0, t0 = po.zhang_init(off, uv, xyz)
    c = pc.Calibrator(1600, 1000)
    c.SetK(K0)
    c.ForceDistortionToConstant(4)               # k3 frozen, as cam_calibration.py:308 does
    c.Optimize(img, world, [q for q in q0], [t for t in t0])
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    io, _, _, so = po.intrinsics_solve(off, uv, xyz, intr0, q0.astype(np.float64), t0.astype(np.float64), const_mask=1 << 8)
    got = np.concatenate([[c.GetK()[0, 0], c.GetK()[1, 1], c.GetK()[0, 2], c.GetK()[1, 2]], c.GetDistortion()]).astype(np.float32)
    want = io.astype(np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and got[8] == 0.0
    assert c.LastIterations() == so["iterations"]


def test_distort_undistort_methods():
    import pycalibrator as pc
    c = pc.Calibrator(1600, 1000)
    c.SetK(po.FIXTURE_K); c.SetDistortion(po.FIXTURE_DIST)
    rng = np.random.default_rng(0)
    xy = rng.uniform(-0.6, 0.4, size=(200, 2)).astype(np.float32)
    uv = np.array(c.Distort(xy))
    assert np.array_equal(uv, po.distort(po.FIXTURE_K, po.FIXTURE_DIST, xy))
    back = np.array(c.Undistort(uv))
    assert np.abs(back - xy).max() < 2e-6


def test_simple_extrinsics_like_the_reference_test(tmp_path, capfd):
    """src/test_extrinsics_calibrator.cpp:40-150 (downsized to 200 frames): build, Serialize,
    Parse into the same object (4 cameras afterwards), Optimize. The reference only prints; here the
    result is compared with the oracle run on the same (quirky) problem."""
    import pycalibrator as pc
    sc = po.rig_scenario(2, 200, 4)
    e = pc.ExtrinsicsCalibrator()
    for c in range(2):
        e.AddCameraTRig(sc["cam_T"][c].reshape(4, 4).T, freeze=(c == 0))
    wid = 0
    for f in range(200):
        e.AddObservationFrame(sc["frame_T"][f].reshape(4, 4).T)
    for f in range(200):
        for p in range(4):
            w = e.AddWorldPoint(f, sc["world_xyz"][wid])
            for c in range(2):
                k = (f * 4 + p) * 2 + c
                assert sc["obs_cam"][k] == c and sc["obs_world"][k] == wid
                e.AddObservation(c, w, sc["obs_uv"][k])
            wid += 1
    fn = str(tmp_path / "serialized_extrinsic_calibration.json")
    e.Serialize(fn)
    e.Parse(fn)
    assert e.NumCameras() == 4                      # the Parse quirk of the reference
    e.Optimize()
    out = capfd.readouterr().out
    assert "iter" in out and e.LastStatus() == 0    # progress goes to stdout like the reference
    # oracle on the identical problem: cameras 0,1 = first (stale) copies, now unfrozen & observed;
    # 2,3 = parsed copies, 2 frozen, never observed
    cam_T = np.concatenate([sc["cam_T"], sc["cam_T"]])
    cq, ct = po.affine_to_qt(cam_T); fq, ft = po.affine_to_qt(sc["frame_T"])
    frozen = np.array([0, 0, 1, 0], dtype=np.uint8)
    r = po.rig_solve(4, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, frozen, fq, ft)
    want = po.qt_to_affine(r[0], r[1])
    for c in range(4):
        got = e.GetCameraTRig(c).T.reshape(-1)
        assert np.abs(got - want[c]).max() < 5e-6
    assert np.array_equal(e.GetCameraTRig(2), sc["cam_T"][0].reshape(4, 4).T)     # unobserved copy untouched up to the float round trip
    cam, idx, wpid, uv, cost = e.GetObservation(3, 5)
    assert np.isclose(cost, r[4][sc["frame_offsets"][3] + 5], rtol=1e-6, atol=1e-15)
    assert e.LastIterations() == r[5]["iterations"]
    assert e.LastSolverReruns() == 0 and e.LastSolverNote() == ""
