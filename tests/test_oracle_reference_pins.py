"""The reference's own tests for this path and its neighbours, re-expressed on the restatement.

src/test_calibrator.cpp:30-75 (generator sizes; K within 1 % after Estimate),
src/test_geometry.cpp:163-194 (FixRotationMatrix validity; EstimateHomography recovers H +-1e-4).
The rig test (src/test_extrinsics_calibrator.cpp) asserts nothing, so there is nothing to pin.
"""
import numpy as np

from oracle import pyoracle as po


def test_data_is_generated():            # test_calibrator.cpp:30-35
    uv, xyz = po.Generator().points(100)
    assert uv.shape == (100, 2) and xyz.shape == (100, 3)
    assert np.all(uv >= 0) and np.all(uv[:, 0] < 1599) and np.all(uv[:, 1] < 999)


def test_planar_data_is_generated():     # test_calibrator.cpp:37-43
    uv, xyz = po.Generator().planar(100)
    assert uv.shape == (100, 2) and xyz.shape == (100, 3)
    assert np.all(xyz[:, 2] == 0)        # data_generator.cpp:119


def test_calibration_estimation_works():  # test_calibrator.cpp:45-75
    off, uv, xyz = po.make_intrinsics_problem(5, 100)
    K0, q0, t0 = po.zhang_init(off, uv, xyz)                      # Calibrator::Estimate, calibrator.cpp:47-66
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    intr, _, _, _ = po.intrinsics_solve(off, uv, xyz, intr0, q0.astype(np.float64), t0.astype(np.float64))
    new_K = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1]], dtype=np.float32)  # :326-330
    K = po.FIXTURE_K
    for i in range(3):
        for j in range(3):
            if K[i, j] != 0:
                assert (new_K[i, j] - K[i, j]) / K[i, j] < 0.01   # the reference's one-sided check
                assert abs(new_K[i, j] - K[i, j]) / K[i, j] < 0.01  # and the two-sided one it meant
            else:
                assert new_K[i, j] == 0


def _valid_rotation(R):
    return (np.allclose(R @ R.T, np.eye(3), atol=1e-5) and np.allclose(np.linalg.norm(R, axis=0), 1, atol=1e-5))


def test_fix_rotation_matrix():          # test_geometry.cpp:66-84,163-170
    rng = np.random.default_rng(0)
    for _ in range(100):
        R = rng.uniform(-1, 1, size=(3, 3)).astype(np.float32)
        assert _valid_rotation(po.fix_rotation_matrix(R).astype(np.float64))


def test_estimate_homography():          # test_geometry.cpp:172-194
    rng = np.random.default_rng(1)
    H = rng.uniform(-1, 1, size=(3, 3)).astype(np.float32)
    p1 = rng.uniform(-1, 1, size=(10, 2)).astype(np.float32)
    ph = np.concatenate([p1, np.ones((10, 1), np.float32)], 1) @ H.T
    p2 = (ph[:, :2] / ph[:, 2:3]).astype(np.float32)
    He = po.estimate_homography(p1, p2)
    He = He / He[2, 2] * H[2, 2]
    assert np.abs(He - H).max() < 1e-4 * max(1.0, np.abs(H).max() / abs(H[2, 2]))


def test_zhang_init_is_close_to_truth():
    off, uv, xyz = po.make_intrinsics_problem(20, 88)
    K, q, t = po.zhang_init(off, uv, xyz)
    assert abs(K[0, 0] - 1000) < 60 and abs(K[1, 1] - 1000) < 60 and abs(K[0, 2] - 800) < 40 and abs(K[1, 2] - 500) < 40
    assert np.all(t[:, 2] > 0) and np.allclose(np.linalg.norm(q, axis=1), 1, atol=1e-5)
