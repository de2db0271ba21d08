"""The identity the rig sweeps rest on, checked on the CPU against the oracle's analytic Jacobians (themselves validated
against the reference functor in tests/test_oracle_model.py and tests/test_oracle_rigk.py): inside a (frame, camera) group
every row's frame columns are its camera columns times ONE 6 x 6 matrix,

    J_frame = J_cam M,   M = | R_c                   0   |    (rows: camera rotation, translation; columns: frame)
                             | 2 [R_c,i x t_f]_i     R_c |    R_c,i = i-th row of R_c, t_f = frame translation

so a group's normal-equation block [cam frame r]^2 is N^T G N with G the Gram of [J_cam r] and N = [I M 0; 0 0 1]
(camera_calibrator_amd/csrc/cc_rig.hip, k_rig_sweep_adj / k_rig_sweep_adjk)."""
import numpy as np
import pytest

from oracle import pyoracle as po


def _quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def _rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _adjoint(q_cr, t_rw):
    Rc = _rot(q_cr)
    M = np.zeros((6, 6))
    M[:3, :3] = Rc
    M[3:, 3:] = Rc
    for i in range(3):
        M[3 + i, :3] = 2.0 * np.cross(Rc[i], t_rw)
    return M


@pytest.mark.parametrize("seed", range(5))
def test_frame_columns_are_the_camera_columns_times_the_group_adjoint(seed):
    rng = np.random.default_rng(seed)
    q_rw, q_cr = _quat(rng), _quat(rng)
    t_rw, t_cr = rng.normal(size=3) * 2.0, rng.normal(size=3) * 0.5
    M = _adjoint(q_cr, t_rw)
    Rf, Rc = _rot(q_rw), _rot(q_cr)
    for _ in range(50):
        X = rng.normal(size=3) * 0.4
        xc = Rc @ (Rf @ X + t_rw) + t_cr
        if xc[2] < 0.2:          # behind or too close to the camera: not a configuration the solver meets
            X[:] = np.linalg.solve(Rc @ Rf, np.array([0.1, -0.2, 3.0]) - t_cr - Rc @ t_rw)
        uv = rng.normal(size=2) * 0.1
        _, J = po.rig_residual(q_rw, t_rw, q_cr, t_cr, X, uv)
        scale = np.abs(J).max()
        assert np.abs(J[:, 6:12] - J[:, 0:6] @ M).max() <= 1e-13 * scale * (1.0 + np.abs(M).max())


def test_group_block_is_the_seven_column_gram_sandwiched_by_n():
    rng = np.random.default_rng(7)
    q_rw, q_cr = _quat(rng), _quat(rng)
    t_rw, t_cr = np.array([0.3, -0.2, 2.5]), np.array([0.1, 0.05, -0.02])
    Rf, Rc = _rot(q_rw), _rot(q_cr)
    rows = []
    for _ in range(200):
        X = np.linalg.solve(Rc @ Rf, np.array([rng.normal() * 0.3, rng.normal() * 0.3, 2.0 + rng.random()]) - t_cr - Rc @ t_rw)
        xc = Rc @ (Rf @ X + t_rw) + t_cr
        uv = xc[:2] / xc[2] + rng.normal(size=2) * 1e-3
        res, J = po.rig_residual(q_rw, t_rw, q_cr, t_cr, X, uv)
        rows.append(np.hstack([J, res[:, None]]))          # [cam 6 | frame 6 | r]
    A = np.vstack(rows)
    full = A.T @ A                                          # what the 13-column product accumulates
    X7 = A[:, [0, 1, 2, 3, 4, 5, 12]]
    G = X7.T @ X7
    N = np.zeros((7, 13))
    N[:6, :6] = np.eye(6)
    N[:6, 6:12] = _adjoint(q_cr, t_rw)
    N[6, 12] = 1.0
    derived = N.T @ G @ N
    assert np.abs(derived - full).max() <= 1e-12 * np.abs(full).max()
    # model-cost term of a step d = [d_c d_f]: q = d^T g + 1/2 d^T H d = 1/2 (e'^T G e' - G[6][6]), e' = [d_c + M d_f, 1]
    d = rng.normal(size=12) * 1e-2
    q_full = d @ full[:12, 12] + 0.5 * d @ full[:12, :12] @ d
    e = np.append(d[:6] + N[:6, 6:12] @ d[6:], 1.0)
    assert np.isclose(q_full, 0.5 * (e @ G @ e - G[6, 6]), rtol=1e-10, atol=1e-14 * np.abs(full).max())


def test_the_identity_carries_over_to_the_pixel_model_of_the_extension():
    rng = np.random.default_rng(11)
    intr = np.array([1000.0, 1005.0, 800.0, 500.0, -0.05, 0.01, 1e-3, -5e-4, 2e-3])
    q_rw, q_cr = _quat(rng), _quat(rng)
    t_rw, t_cr = np.array([-0.4, 0.1, 3.0]), np.array([0.2, -0.1, 0.05])
    Rf, Rc = _rot(q_rw), _rot(q_cr)
    M = _adjoint(q_cr, t_rw)
    for _ in range(50):
        X = np.linalg.solve(Rc @ Rf, np.array([rng.normal() * 0.4, rng.normal() * 0.4, 2.0 + rng.random()]) - t_cr - Rc @ t_rw)
        uv = np.array([800.0, 500.0]) + rng.normal(size=2) * 50.0
        _, J = po.rigk_residual(intr, q_rw, t_rw, q_cr, t_cr, X, uv)     # [cam 6 | frame 6 | intrinsics 9]
        assert np.abs(J[:, 6:12] - J[:, 0:6] @ M).max() <= 1e-13 * np.abs(J[:, :12]).max() * (1.0 + np.abs(M).max())
