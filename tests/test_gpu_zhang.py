"""Zhang initialisation on the device (cc_zhang_init) against the oracle's SVD-based restatement of
geometry.cpp:70-203 / calibrator.cpp:47-66. Tolerances: homographies (normalised, sign-aligned)
1e-5 relative -- float32 storage, and 1e-4 is what the reference's own test allows
(test_geometry.cpp:172-194); K 1e-5 relative; poses 2e-5 absolute (float32 outputs)."""
import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frames,pts", [(3, 4), (5, 100), (20, 88), (12, [4, 9, 64, 65, 300, 5, 257, 128, 77, 500, 31, 1000])])
def test_zhang_init_matches_oracle(frames, pts):
    off, uv, xyz = po.make_intrinsics_problem(frames, pts)
    Kg, qg, tg, Hg = capi.zhang_init(off, uv, xyz, want_homographies=True)
    Ko, qo, to = po.zhang_init(off, uv, xyz)
    for f in range(frames):
        Ho = po.estimate_homography(xyz[off[f]:off[f + 1]], uv[off[f]:off[f + 1]])
        a, b = Hg[f] / np.linalg.norm(Hg[f]), Ho / np.linalg.norm(Ho)
        if np.sum(a * b) < 0:
            b = -b
        assert np.abs(a - b).max() < 1e-5
    if frames == 3 and pts == 4:
        return  # K from 3 noisy 4-point homographies is ill-conditioned: only H is compared
    assert np.allclose(Kg, Ko, rtol=1e-5, atol=1e-6)
    assert np.all(tg[:, 2] > 0)
    sign = np.sign(np.sum(qg * qo, axis=1, keepdims=True))
    assert np.abs(qg - sign * qo).max() < 2e-5 and np.abs(tg - to).max() < 2e-5


def test_zhang_init_feeds_the_solver_to_the_same_minimum():
    from tests.helpers import TIGHT
    off, uv, xyz = po.make_intrinsics_problem(20, 88)
    Kg, qg, tg = capi.zhang_init(off, uv, xyz)
    intr0 = np.array([Kg[0, 0], Kg[1, 1], Kg[0, 2], Kg[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    ig, _, _, sg = capi.intrinsics_optimize(off, uv, xyz, intr0, qg.astype(np.float64), tg.astype(np.float64),
                                            options=capi.default_options(**TIGHT))
    Ko, qo, to = po.zhang_init(off, uv, xyz)
    intr1 = np.array([Ko[0, 0], Ko[1, 1], Ko[0, 2], Ko[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    io, _, _, so = po.intrinsics_solve(off, uv, xyz, intr1, qo.astype(np.float64), to.astype(np.float64),
                                       options=po.default_options(**TIGHT))
    assert np.all(np.abs(ig[:4] - io[:4]) <= 1e-9 * np.abs(io[:4])) and np.all(np.abs(ig[4:] - io[4:]) <= 1e-9)


def test_zhang_init_at_full_size_with_badly_graded_frames():
    """BASELINE.json configs[2] size: some of the 1000 random planes are nearly edge-on (world coordinates of 1e5,
    DLT singular values spread over 1.5e7): the normal-equation route must still deliver a homography of float32
    quality for every frame, and the solver must start from it as it does from the SVD route."""
    off, uv, xyz = po.make_intrinsics_problem(1000, 500)
    Kg, qg, tg, Hg = capi.zhang_init(off, uv, xyz, want_homographies=True)
    Ko, qo, to = po.zhang_init(off, uv, xyz)
    assert np.allclose(Kg, Ko, rtol=2e-5, atol=1e-4)
    sign = np.sign(np.sum(qg * qo, axis=1, keepdims=True))
    assert np.abs(qg - sign * qo).max() < 2e-4 and np.abs(tg - to).max() < 2e-4
    for f in (875, 0, 500, 999):
        Ho = po.estimate_homography(xyz[off[f]:off[f + 1]], uv[off[f]:off[f + 1]])
        a, b = Hg[f] / np.linalg.norm(Hg[f]), Ho / np.linalg.norm(Ho)
        if np.sum(a * b) < 0:
            b = -b
        assert np.abs(a - b).max() < 2e-5
    intr0 = np.array([Kg[0, 0], Kg[1, 1], Kg[0, 2], Kg[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    r = capi.intrinsics_optimize(off, uv, xyz, intr0, qg.astype(np.float64), tg.astype(np.float64))
    assert r[3]["iterations"] == 4 and r[3]["termination"] == "FUNCTION"


def test_zhang_init_rejects_bad_input():
    off, uv, xyz = po.make_intrinsics_problem(2, 10)
    with pytest.raises(capi.CcError):
        capi.zhang_init(off, uv, xyz)          # needs >= 3 frames (geometry.cpp:126)
    off, uv, xyz = po.make_intrinsics_problem(3, [10, 3, 10])
    with pytest.raises(capi.CcError):
        capi.zhang_init(off, uv, xyz)          # a homography needs >= 4 points


@pytest.mark.gpu
@pytest.mark.parametrize("frames,pts,mask", [(20, 88, 0), (200, 200, 1 << 8)])
def test_estimate_in_one_call_equals_initialisation_plus_optimize(frames, pts, mask):
    """cc_intrinsics_estimate (what Calibrator::Estimate calls on one device) = cc_zhang_init followed by
    cc_intrinsics_optimize from its float K and poses: same bits, one upload."""
    off, uv, xyz = capi.make_intrinsics_problem(frames, pts)
    K0, q0, t0 = capi.zhang_init(off, uv, xyz)
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    i2, q2, t2, s2 = capi.intrinsics_optimize(off, uv, xyz, intr0, q0.astype(np.float64), t0.astype(np.float64), const_mask=mask)
    K1, i1, q1, t1, s1 = capi.intrinsics_estimate(off, uv, xyz, const_mask=mask)
    assert np.array_equal(K1, K0)
    assert s1["iterations"] == s2["iterations"] and s1["termination"] == s2["termination"]
    assert np.array_equal(i1, i2) and np.array_equal(q1, q2) and np.array_equal(t1, t2)
    assert s1["final_cost"] == s2["final_cost"]


@pytest.mark.gpu
@pytest.mark.parametrize("frames,pts", [(20, 88), (7, [30, 4, 500, 12, 64, 65, 9]), (400, 300)])
def test_views_entry_points_equal_the_flat_ones(frames, pts):
    """cc_intrinsics_estimate_views / cc_intrinsics_optimize_views (views as separate arrays, packed by the library piece by
    piece under its upload -- three pieces at 400 x 300) give the bits of the flat entry points."""
    off, uv, xyz = capi.make_intrinsics_problem(frames, pts)
    K0, i0, q0, t0, s0 = capi.intrinsics_estimate(off, uv, xyz)
    K1, i1, q1, t1, s1 = capi.intrinsics_estimate(off, uv, xyz, views=True)
    assert np.array_equal(K0, K1) and np.array_equal(i0, i1) and np.array_equal(q0, q1) and np.array_equal(t0, t1)
    assert s0["iterations"] == s1["iterations"] and s0["final_cost"] == s1["final_cost"]
    Kz, qz, tz = capi.zhang_init(off, uv, xyz)
    intr = np.array([Kz[0, 0], Kz[1, 1], Kz[0, 2], Kz[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    a = capi.intrinsics_optimize(off, uv, xyz, intr, qz.astype(np.float64), tz.astype(np.float64))
    b = capi.intrinsics_optimize(off, uv, xyz, intr, qz.astype(np.float64), tz.astype(np.float64), views=True)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert a[3]["final_cost"] == b[3]["final_cost"]


@pytest.mark.gpu
def test_views_entry_points_refuse_short_views():
    off, uv, xyz = capi.make_intrinsics_problem(4, [10, 3, 10, 10])
    with pytest.raises(capi.CcError):
        capi.intrinsics_estimate(off, uv, xyz, views=True)   # a homography needs >= 4 points
