"""Shared helpers for the test-suite (problem builders on top of the oracle's generator)."""
import numpy as np

from oracle import pyoracle as po


def intrinsics_case(n_frames, pts, **gen_kw):
    """Synthetic single-camera problem + Calibrator::Estimate-style initial state."""
    off, uv, xyz = po.make_intrinsics_problem(n_frames, pts, **gen_kw)
    K, q, t = po.zhang_init(off, uv, xyz)
    intr0 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    return dict(off=off, uv=uv, xyz=xyz, intr0=intr0, q0=q.astype(np.float64), t0=t.astype(np.float64))


def quat_plus(q, d):
    """ceres::QuaternionManifold::Plus, numpy restatement used by finite-difference checks."""
    q = np.asarray(q, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    nd = np.linalg.norm(d)
    if nd == 0:
        return q.copy()
    a = np.concatenate([[np.cos(nd)], np.sin(nd) / nd * d])
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = q
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def block_rel_err(a, b):
    scale = np.abs(b).max(axis=(1, 2), keepdims=True)
    scale[scale == 0] = 1.0
    return float((np.abs(a - b) / scale).max())


TIGHT = dict(function_tolerance=1e-15, gradient_tolerance=1e-13, parameter_tolerance=1e-14, max_iterations=60)
