"""The oracle's residual models and analytic Jacobians against independent derivations.

(i) sympy symbolic differentiation of the model exactly as written in the reference
    (calibrator.cpp:70-95,199-214; extrinsics_calibrator.cpp:57-80), evaluated with 40-digit mpmath;
(ii) central finite differences through ceres::QuaternionManifold::Plus.
"""
import mpmath as mp
import numpy as np
import pytest
import sympy as sp

from oracle import pyoracle as po
from tests.helpers import quat_plus


def _rot(q):
    w, x, y, z = q
    n = sp.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return sp.Matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _qmul(a, b):
    return [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]


def _plus_first_order(q, d):
    # QuaternionManifold::Plus linearised at delta = 0: [cos|d|, sin|d|/|d| d] -> [1, d]
    return _qmul([sp.Integer(1), d[0], d[1], d[2]], q)


@pytest.fixture(scope="module")
def intr_symbolic():
    k = sp.symbols("fx fy px py k1 k2 p1 p2 k3")
    q = sp.symbols("qw qx qy qz")
    t = sp.symbols("tx ty tz")
    X = sp.symbols("X Y Z")
    uv = sp.symbols("u v")
    d = sp.symbols("d1 d2 d3")
    fx, fy, px, py, k1, k2, p1, p2, k3 = k
    xc = _rot(_plus_first_order(list(q), d)) * sp.Matrix(X) + sp.Matrix(t)
    xn, yn = xc[0] / xc[2], xc[1] / xc[2]
    r2 = xn * xn + yn * yn
    r_mult = 1 + k1 * r2 + k2 * r2**2 + k3 * r2**3                        # calibrator.cpp:77-80
    nx = xn * r_mult + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)          # :81
    ny = yn * r_mult + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)          # :82
    res = sp.Matrix([fx * nx + px - uv[0], fy * ny + py - uv[1]])          # :93-94,213-214
    params = list(k) + list(d) + list(t)
    J = res.jacobian(params).subs({d[0]: 0, d[1]: 0, d[2]: 0})
    res0 = res.subs({d[0]: 0, d[1]: 0, d[2]: 0})
    args = list(k) + list(q) + list(t) + list(X) + list(uv)
    return sp.lambdify(args, [res0, J], modules="mpmath")


def test_intrinsics_jacobian_matches_sympy(intr_symbolic):
    mp.mp.dps = 40
    rng = np.random.default_rng(3)
    for _ in range(12):
        intr = np.array([1000, 990, 800, 500, -4e-2, 5e-4, 1e-3, 2e-5, -3e-4]) * (1 + 0.1 * rng.normal(size=9))
        q = rng.normal(size=4)                      # deliberately not unit-norm
        t = np.array([0.05, -0.03, 0.6]) + 0.05 * rng.normal(size=3)
        X = 0.3 * rng.normal(size=3)
        uv = np.array([700.0, 400.0]) + 50 * rng.normal(size=2)
        res, J = po.intrinsics_residual(intr, q, t, X, uv)
        args = [mp.mpf(float(v)) for v in np.concatenate([intr, q, t, X, uv])]
        r_s, J_s = intr_symbolic(*args)
        r_s = np.array([float(v) for v in r_s], dtype=np.float64)
        J_s = np.array([[float(J_s[i, j]) for j in range(15)] for i in range(2)])
        assert np.allclose(res, r_s, rtol=1e-12, atol=1e-10)
        scale = np.abs(J_s).max(axis=0, keepdims=True) + 1e-300
        assert (np.abs(J - J_s) / scale).max() < 1e-12


def test_rig_jacobian_matches_sympy():
    mp.mp.dps = 40
    qf = sp.symbols("aw ax ay az"); tf = sp.symbols("fx_ fy_ fz_")
    qc = sp.symbols("cw cx cy cz"); tc = sp.symbols("gx gy gz")
    X = sp.symbols("X Y Z"); uv = sp.symbols("u v")
    dc = sp.symbols("dc1 dc2 dc3"); df = sp.symbols("df1 df2 df3")
    x_rig = _rot(_plus_first_order(list(qf), df)) * sp.Matrix(X) + sp.Matrix(tf)   # extrinsics_calibrator.cpp:61-65
    x = _rot(_plus_first_order(list(qc), dc)) * x_rig + sp.Matrix(tc)              # :68-72
    res = sp.Matrix([x[0] / x[2] - uv[0], x[1] / x[2] - uv[1]])                    # :75-79
    params = list(dc) + list(tc) + list(df) + list(tf)
    zero = {s: 0 for s in list(dc) + list(df)}
    J = res.jacobian(params).subs(zero)
    fn = sp.lambdify(list(qf) + list(tf) + list(qc) + list(tc) + list(X) + list(uv), [res.subs(zero), J], modules="mpmath")
    rng = np.random.default_rng(5)
    for _ in range(8):
        qrw = rng.normal(size=4); trw = np.array([0.5, 0.4, 0.7]) + 0.1 * rng.normal(size=3)
        qcr = np.array([1, 0.01, -0.02, 0.03]) + 0.01 * rng.normal(size=4); tcr = 0.03 * rng.normal(size=3)
        Xw = 0.2 * rng.normal(size=3); m = 0.2 * rng.normal(size=2)
        res_o, J_o = po.rig_residual(qrw, trw, qcr, tcr, Xw, m)
        r_s, J_s = fn(*[mp.mpf(float(v)) for v in np.concatenate([qrw, trw, qcr, tcr, Xw, m])])
        J_s = np.array([[float(J_s[i, j]) for j in range(12)] for i in range(2)])
        assert np.allclose(res_o, [float(v) for v in r_s], rtol=1e-12, atol=1e-13)
        assert np.abs(J_o - J_s).max() < 1e-11 * max(1.0, np.abs(J_s).max())


def test_intrinsics_jacobian_finite_differences_through_manifold_plus():
    rng = np.random.default_rng(1)
    intr = np.array([1000, 990, 800, 500, -4e-2, 5e-4, 1e-3, 2e-5, -3e-4])
    q = rng.normal(size=4); t = np.array([0.05, -0.03, 0.6]); X = np.array([0.1, -0.2, 0.05]); uv = np.array([700.0, 400.0])
    _, J = po.intrinsics_residual(intr, q, t, X, uv)

    def f(d):
        return po.intrinsics_residual(intr + d[:9], quat_plus(q, d[9:12]), t + d[12:15], X, uv, False)[0]

    Jn = np.zeros((2, 15))
    for i in range(15):
        h = 1e-6 * max(1.0, abs(intr[i]) if i < 9 else 1.0)
        e = np.zeros(15); e[i] = h
        Jn[:, i] = (f(e) - f(-e)) / (2 * h)
    assert np.abs(J - Jn).max() / np.abs(J).max() < 1e-8


def test_residual_is_invariant_to_quaternion_scale_and_sign():
    rng = np.random.default_rng(2)
    intr = np.array([1000, 990, 800, 500, -4e-2, 5e-4, 1e-3, 2e-5, -3e-4])
    q = rng.normal(size=4); t = np.array([0.05, -0.03, 0.6]); X = np.array([0.1, -0.2, 0.05]); uv = np.array([700.0, 400.0])
    r0, J0 = po.intrinsics_residual(intr, q, t, X, uv)
    for s in (-1.0, 3.7):
        r1, J1 = po.intrinsics_residual(intr, s * q, t, X, uv)   # QuaternionRotatePoint normalises (calibrator.cpp:201)
        assert np.allclose(r0, r1, rtol=1e-13, atol=1e-11) and np.allclose(J0, J1, rtol=1e-11, atol=1e-9)


def test_distort_matches_float_restatement_and_undistort_inverts():
    rng = np.random.default_rng(0)
    xy = rng.uniform(-0.7, 0.45, size=(2000, 2)).astype(np.float32)
    uv = po.distort(po.FIXTURE_K, po.FIXTURE_DIST, xy)
    # double-precision model, calibrator.cpp:70-95
    k1, k2, p1, p2, k3 = [float(v) for v in po.FIXTURE_DIST]
    x, y = xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64)
    r2 = x * x + y * y
    m = 1 + k1 * r2 + k2 * r2**2 + k3 * r2**3
    u = 1000 * (x * m + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)) + 800
    v = 1000 * (y * m + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)) + 500
    assert np.abs(uv[:, 0] - u).max() < 2e-4 and np.abs(uv[:, 1] - v).max() < 2e-4   # float32 pixels
    back = po.undistort(po.FIXTURE_K, po.FIXTURE_DIST, uv)
    assert np.abs(back - xy).max() < 2e-6
