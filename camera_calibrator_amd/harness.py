"""Synthetic rig scenarios for the EXTENSION problems (rig poses + intrinsics on pixel observations; nothing in the
reference generates such data). Pure numpy, host code: bench.py, the scripts and the tests build inputs with it; the
solver never calls it. The reference's own scenarios live behind include/cc_harness.h (capi.make_intrinsics_problem,
capi.rig_scenario)."""
import numpy as np


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


RIGK_INTR_TRUE = np.array([1000.0, 1000.0, 800.0, 500.0, -4.0e-2, 5e-4, 1.0e-3, 2.0e-5, -3e-4])  # test_calibrator.cpp:14-19


def _quat_to_R(q):
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rigk_case(cams, frames, pts, seed=0, pixel_noise=0.5, per_camera=False):
    """EXTENSION scenario: a rig of `cams` cameras (camera 0 = rig frame, frozen; the others
    offset by U(-0.03, 0.03) in x, y as in test_extrinsics_calibrator.cpp:60-69) watching, in every frame, `pts`
    world points of a +-0.2 cube from 0.6-1.2 m, every camera seeing every point; pixel observations through the
    fixture camera of test_calibrator.cpp:14-19 with U(-noise, noise) px. Initial state: cameras off by 5 mm /
    0.1 deg, frames by 20 mm / 1 deg (the magnitudes of test_extrinsics_calibrator.cpp:53-58), focal lengths
    2 % off, principal point 5 px off, no distortion.
    per_camera=True: every camera has a camera of its own (focal lengths and principal point a few per cent /
    pixels apart, distortion scaled), returned as intr_true [cams, 9]; intr0 is then [cams, 9] too."""
    rng = np.random.default_rng(seed)
    ident = np.array([1.0, 0, 0, 0])

    def small_rot(deg):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        return axis * np.deg2rad(rng.uniform(-deg, deg)) / 2.0   # Plus() rotates by twice |delta|

    cam_q = np.tile(ident, (cams, 1))
    cam_t = np.zeros((cams, 3))
    cam_t[1:, :2] = rng.uniform(-0.03, 0.03, size=(cams - 1, 2))
    cam_q0, cam_t0 = cam_q.copy(), cam_t.copy()
    for c in range(1, cams):
        cam_q0[c] = quat_plus(cam_q[c], small_rot(0.1))
        cam_t0[c] += rng.uniform(-0.005, 0.005, size=3)
    frame_q = np.zeros((frames, 4))
    frame_t = np.zeros((frames, 3))
    frame_q0, frame_t0 = np.zeros((frames, 4)), np.zeros((frames, 3))
    world = np.zeros((frames * pts, 3), dtype=np.float32)
    obs_cam, obs_world, obs_uv = [], [], []
    k_all = np.tile(RIGK_INTR_TRUE, (cams, 1))
    if per_camera:
        krng = np.random.default_rng(seed + 12345)   # its own stream: the rest of the scenario does not change
        k_all[:, :2] *= 1 + krng.uniform(-0.03, 0.03, size=(cams, 2))
        k_all[:, 2:4] += krng.uniform(-15, 15, size=(cams, 2))
        k_all[:, 4:] *= krng.uniform(0.7, 1.3, size=(cams, 1))
    for f in range(frames):
        frame_q[f] = quat_plus(ident, small_rot(15.0))
        frame_t[f] = [rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.6, 1.2)]
        frame_q0[f] = quat_plus(frame_q[f], small_rot(1.0))
        frame_t0[f] = frame_t[f] + rng.uniform(-0.02, 0.02, size=3)
        Rf = _quat_to_R(frame_q[f])
        X = rng.uniform(-0.2, 0.2, size=(pts, 3))
        world[f * pts:(f + 1) * pts] = (X + rng.uniform(-0.001, 0.001, size=(pts, 3))).astype(np.float32)
        Xr = X @ Rf.T + frame_t[f]
        for c in range(cams):
            k = k_all[c]
            Xc = Xr @ _quat_to_R(cam_q[c]).T + cam_t[c]
            x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
            r2 = x * x + y * y
            m = 1 + k[4] * r2 + k[5] * r2 ** 2 + k[8] * r2 ** 3
            xd = x * m + 2 * k[6] * x * y + k[7] * (r2 + 2 * x * x)
            yd = y * m + 2 * k[7] * x * y + k[6] * (r2 + 2 * y * y)
            uv = np.stack([k[0] * xd + k[2], k[1] * yd + k[3]], axis=1) + rng.uniform(-pixel_noise, pixel_noise, size=(pts, 2))
            obs_cam.append(np.full(pts, c, dtype=np.uint32))
            obs_world.append(np.arange(f * pts, (f + 1) * pts, dtype=np.uint64))
            obs_uv.append(uv.astype(np.float32))
    # observations of a frame are stored point-major like the reference's AddObservation order
    oc = np.concatenate(obs_cam).reshape(frames, cams, pts).transpose(0, 2, 1).reshape(-1)
    ow = np.concatenate(obs_world).reshape(frames, cams, pts).transpose(0, 2, 1).reshape(-1)
    ou = np.concatenate(obs_uv).reshape(frames, cams, pts, 2).transpose(0, 2, 1, 3).reshape(-1, 2)
    frozen = np.zeros(cams, dtype=np.uint8)
    frozen[0] = 1
    return dict(cams=cams, frame_offsets=(np.arange(frames + 1) * cams * pts).astype(np.int64), obs_cam=np.ascontiguousarray(oc),
                obs_world=np.ascontiguousarray(ow), obs_uv_pix=np.ascontiguousarray(ou), world_xyz=world, cam_frozen=frozen,
                cam_q_true=cam_q, cam_t_true=cam_t, frame_q_true=frame_q, frame_t_true=frame_t,
                cam_q0=cam_q0, cam_t0=cam_t0, frame_q0=frame_q0, frame_t0=frame_t0,
                intr0=(np.tile(np.array([1020.0, 980.0, 805.0, 495.0, 0, 0, 0, 0, 0]), (cams, 1)) if per_camera
                       else np.array([1020.0, 980.0, 805.0, 495.0, 0, 0, 0, 0, 0], dtype=np.float64)),
                intr_true=k_all if per_camera else RIGK_INTR_TRUE)
