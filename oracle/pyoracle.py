"""ctypes binding of oracle/liboracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. Nothing under camera_calibrator_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class Options(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("use_nonmonotonic_steps", C.c_int32),
        ("max_consecutive_nonmonotonic_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("max_consecutive_invalid_steps", C.c_int32),
        ("num_threads", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("model_cost_change", C.c_double),
        ("relative_decrease", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("radius", C.c_double),
        ("accepted", C.c_int32),
        ("valid", C.c_int32),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("successful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("log_len", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("seconds", C.c_double),
        ("log", C.POINTER(Iteration)),
        ("log_capacity", C.c_int32),
        ("pad_", C.c_int32),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32)

TERMINATION = {0: "NO_CONVERGENCE", 1: "GRADIENT", 2: "PARAMETER", 3: "FUNCTION",
               4: "FAILURE_INVALID_STEPS", 5: "MIN_RADIUS"}


def build(force=False):
    src = os.path.join(_HERE, "oracle.cpp")
    hdr = os.path.join(_HERE, "oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


_lib = None
_FAST_PATH = os.path.join(_HERE, "liboracle_fast.so")


def _declare(l):
    l.oc_intrinsics_blocks.restype = C.c_double
    l.oc_generator_create.restype = C.c_void_p
    l.oc_generator_planar.restype = C.c_int64
    l.oc_generator_points.restype = C.c_int64
    return l


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


def use_fast_build():
    """TIMING ONLY (bench.py's cpu_baseline leg): rebuild the oracle on THIS host with -O3 -march=native
     as liboracle_fast.so and route every call of this module through it from now on. Parity
    tests never call this: their library is the -ffp-contract=off build. Returns (flags, cpu model name)."""
    global _lib
    subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "liboracle_fast.so"])
    _lib = _declare(C.CDLL(_FAST_PATH))
    flags = subprocess.check_output(["make", "-s", "-C", _HERE, "--eval", "pf: ; @echo $(CXX) $(FASTFLAGS)", "pf"]).decode().strip()
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return flags, model


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_options(**kw):
    o = Options()
    lib().oc_options_init(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _summary(log_capacity):
    log = (Iteration * max(1, log_capacity))()
    s = Summary()
    s.log = C.cast(log, C.POINTER(Iteration))
    s.log_capacity = log_capacity
    return s, log


def summary_to_dict(s, log):
    n = s.log_len
    names = [f[0] for f in Iteration._fields_]
    return {
        "iterations": s.iterations, "successful_steps": s.successful_steps,
        "termination": TERMINATION.get(s.termination, str(s.termination)),
        "initial_cost": s.initial_cost, "final_cost": s.final_cost, "seconds": s.seconds,
        "log": [{k: getattr(log[i], k) for k in names} for i in range(n)],
    }


# ---------------------------------------------------------------------------------------------
def intrinsics_residual(intr, q, t, X, uv, want_jacobian=True):
    res = np.zeros(2)
    J = np.zeros((2, 15)) if want_jacobian else None
    lib().oc_intrinsics_residual(_p(_f64(intr), C.c_double), _p(_f64(q), C.c_double),
                                 _p(_f64(t), C.c_double), _p(_f64(X), C.c_double),
                                 _p(_f64(uv), C.c_double), _p(res, C.c_double),
                                 _p(J, C.c_double) if J is not None else None)
    return res, J


def rig_residual(q_rw, t_rw, q_cr, t_cr, X, uv, want_jacobian=True):
    res = np.zeros(2)
    J = np.zeros((2, 12)) if want_jacobian else None
    lib().oc_rig_residual(_p(_f64(q_rw), C.c_double), _p(_f64(t_rw), C.c_double),
                          _p(_f64(q_cr), C.c_double), _p(_f64(t_cr), C.c_double),
                          _p(_f64(X), C.c_double), _p(_f64(uv), C.c_double), _p(res, C.c_double),
                          _p(J, C.c_double) if J is not None else None)
    return res, J


def rigk_residual(intr, q_rw, t_rw, q_cr, t_cr, X, uv, want_jacobian=True):
    """EXTENSION: rig model composed with the pixel model; J is 2x21 (cam 6, frame 6, intrinsics 9)."""
    res = np.zeros(2)
    J = np.zeros((2, 21)) if want_jacobian else None
    lib().oc_rigk_residual(_p(_f64(intr), C.c_double), _p(_f64(q_rw), C.c_double), _p(_f64(t_rw), C.c_double),
                           _p(_f64(q_cr), C.c_double), _p(_f64(t_cr), C.c_double),
                           _p(_f64(X), C.c_double), _p(_f64(uv), C.c_double), _p(res, C.c_double),
                           _p(J, C.c_double) if J is not None else None)
    return res, J


def rigk_solve(n_cams, frame_offsets, obs_cam, obs_world, obs_uv_pixels, world_xyz, intr, cam_q, cam_t,
               cam_frozen, frame_q, frame_t, const_mask=0, huber_a=0.0, options=None, log_capacity=2048):
    """EXTENSION: rig poses + shared intrinsics. Returns (intr, cam_q, cam_t, frame_q, frame_t, obs_cost, summary)."""
    offs = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(offs) - 1
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.uint32)
    obs_world = np.ascontiguousarray(obs_world, dtype=np.uint64)
    obs_uv, world_xyz = _f32(obs_uv_pixels), _f32(world_xyz)
    intr = _f64(intr).copy()
    cam_q, cam_t = _f64(cam_q).copy(), _f64(cam_t).copy()
    frame_q, frame_t = _f64(frame_q).copy(), _f64(frame_t).copy()
    frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
    opt = options if options is not None else default_options(max_iterations=1000)
    s, log = _summary(log_capacity)
    cost = np.zeros(len(obs_cam))
    n_world = C.c_int64(len(world_xyz) // 3 if world_xyz.ndim == 1 else world_xyz.shape[0])
    rc = lib().oc_rigk_solve(C.byref(opt), C.c_int64(n_cams), C.c_int64(F), n_world, _p(offs, C.c_int64),
                             _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64), _p(obs_uv, C.c_float),
                             _p(world_xyz, C.c_float), _p(intr, C.c_double), C.c_uint32(const_mask),
                             _p(cam_q, C.c_double), _p(cam_t, C.c_double), _p(frozen, C.c_uint8),
                             _p(frame_q, C.c_double), _p(frame_t, C.c_double), C.c_double(huber_a),
                             _p(cost, C.c_double), C.byref(s))
    if rc != 0:
        raise RuntimeError(f"oc_rigk_solve failed: {rc}")
    return intr, cam_q, cam_t, frame_q, frame_t, cost, summary_to_dict(s, log)


def rigk_solve_per_camera(n_cams, frame_offsets, obs_cam, obs_world, obs_uv_pixels, world_xyz, intr, cam_q, cam_t,
                          cam_frozen, frame_q, frame_t, const_masks=None, huber_a=0.0, options=None, log_capacity=2048):
    """EXTENSION: rig poses + one set of 9 intrinsics per camera (intr: [n_cams, 9], const_masks: [n_cams]).
    Returns (intr [n_cams, 9], cam_q, cam_t, frame_q, frame_t, obs_cost, summary)."""
    offs = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(offs) - 1
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.uint32)
    obs_world = np.ascontiguousarray(obs_world, dtype=np.uint64)
    obs_uv, world_xyz = _f32(obs_uv_pixels), _f32(world_xyz)
    intr = _f64(intr).reshape(n_cams, 9).copy()
    masks = np.zeros(n_cams, dtype=np.uint32) if const_masks is None else np.ascontiguousarray(const_masks, dtype=np.uint32)
    cam_q, cam_t = _f64(cam_q).copy(), _f64(cam_t).copy()
    frame_q, frame_t = _f64(frame_q).copy(), _f64(frame_t).copy()
    frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
    opt = options if options is not None else default_options(max_iterations=1000)
    s, log = _summary(log_capacity)
    cost = np.zeros(len(obs_cam))
    n_world = C.c_int64(len(world_xyz) // 3 if world_xyz.ndim == 1 else world_xyz.shape[0])
    rc = lib().oc_rigk_solve_sets(C.byref(opt), C.c_int64(n_cams), C.c_int64(F), n_world, _p(offs, C.c_int64),
                                  _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64), _p(obs_uv, C.c_float),
                                  _p(world_xyz, C.c_float), C.c_int32(1), _p(intr, C.c_double), _p(masks, C.c_uint32),
                                  _p(cam_q, C.c_double), _p(cam_t, C.c_double), _p(frozen, C.c_uint8),
                                  _p(frame_q, C.c_double), _p(frame_t, C.c_double), C.c_double(huber_a),
                                  _p(cost, C.c_double), C.byref(s))
    if rc != 0:
        raise RuntimeError(f"oc_rigk_solve_sets failed: {rc}")
    return intr, cam_q, cam_t, frame_q, frame_t, cost, summary_to_dict(s, log)


def intrinsics_blocks(offsets, uv, xyz, intr, q, t, const_mask=0, want_blocks=True, num_threads=1):
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    F = len(offsets) - 1
    uv, xyz, intr, q, t = _f32(uv), _f32(xyz), _f64(intr), _f64(q), _f64(t)
    blocks = np.zeros((F, 16, 16)) if want_blocks else None
    cost = lib().oc_intrinsics_blocks(C.c_int64(F), _p(offsets, C.c_int64), _p(uv, C.c_float),
                                      _p(xyz, C.c_float), _p(intr, C.c_double),
                                      C.c_uint32(const_mask), _p(q, C.c_double), _p(t, C.c_double),
                                      _p(blocks, C.c_double) if want_blocks else None,
                                      C.c_int32(num_threads))
    return cost, blocks


def intrinsics_solve(offsets, uv, xyz, intr, q, t, const_mask=0, options=None, log_capacity=1024,
                     allreduce=None):
    """Returns (intr, q, t, summary_dict); inputs are not modified."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    F = len(offsets) - 1
    uv, xyz = _f32(uv), _f32(xyz)
    intr, q, t = _f64(intr).copy(), _f64(q).copy(), _f64(t).copy()
    opt = options if options is not None else default_options()
    s, log = _summary(log_capacity)
    args = [C.byref(opt), C.c_int64(F), _p(offsets, C.c_int64), _p(uv, C.c_float),
            _p(xyz, C.c_float), _p(intr, C.c_double), C.c_uint32(const_mask), _p(q, C.c_double),
            _p(t, C.c_double), C.byref(s)]
    if allreduce is None:
        rc = lib().oc_intrinsics_solve(*args)
    else:
        cb = ALLREDUCE_FN(allreduce)
        rc = lib().oc_intrinsics_solve_sharded(*args, cb, None)
    assert rc == 0
    return intr, q, t, summary_to_dict(s, log)


def rig_solve(n_cams, frame_offsets, obs_cam, obs_world, obs_uv, world_xyz, cam_q, cam_t,
              cam_frozen, frame_q, frame_t, huber_a=None, options=None, log_capacity=2048,
              allreduce=None, cam_seen_global=None):
    offs = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(offs) - 1
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.uint32)
    obs_world = np.ascontiguousarray(obs_world, dtype=np.uint64)
    obs_uv, world_xyz = _f32(obs_uv), _f32(world_xyz)
    cam_q, cam_t = _f64(cam_q).copy(), _f64(cam_t).copy()
    frame_q, frame_t = _f64(frame_q).copy(), _f64(frame_t).copy()
    frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
    if huber_a is None:
        huber_a = float(np.float32(3.0) / np.float32(500.0))  # extrinsics_calibrator.cpp:176
    opt = options if options is not None else default_options(max_iterations=1000)
    s, log = _summary(log_capacity)
    n_obs = len(obs_cam)
    cost = np.zeros(n_obs)
    n_world = C.c_int64(len(world_xyz) // 3 if world_xyz.ndim == 1 else world_xyz.shape[0])
    if allreduce is None:
        rc = lib().oc_rig_solve(C.byref(opt), C.c_int64(n_cams), C.c_int64(F), n_world,
                                _p(offs, C.c_int64), _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64),
                                _p(obs_uv, C.c_float), _p(world_xyz, C.c_float), _p(cam_q, C.c_double),
                                _p(cam_t, C.c_double), _p(frozen, C.c_uint8), _p(frame_q, C.c_double),
                                _p(frame_t, C.c_double), C.c_double(huber_a), _p(cost, C.c_double),
                                C.byref(s))
    else:
        cb = ALLREDUCE_FN(allreduce)
        seen = np.ascontiguousarray(cam_seen_global, dtype=np.uint8)
        rc = lib().oc_rig_solve_sharded(C.byref(opt), C.c_int64(n_cams), C.c_int64(F), n_world,
                                        _p(offs, C.c_int64), _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64),
                                        _p(obs_uv, C.c_float), _p(world_xyz, C.c_float), _p(cam_q, C.c_double),
                                        _p(cam_t, C.c_double), _p(frozen, C.c_uint8), _p(seen, C.c_uint8),
                                        _p(frame_q, C.c_double), _p(frame_t, C.c_double), C.c_double(huber_a),
                                        _p(cost, C.c_double), C.byref(s), cb, None)
    assert rc == 0
    return cam_q, cam_t, frame_q, frame_t, cost, summary_to_dict(s, log)


def distort(K, dist, xy):
    xy = _f32(xy)
    out = np.zeros_like(xy)
    lib().oc_distort(_p(_f32(K), C.c_float), _p(_f32(dist), C.c_float), C.c_int64(xy.size // 2),
                     _p(xy, C.c_float), _p(out, C.c_float))
    return out


def undistort(K, dist, uv):
    uv = _f32(uv)
    out = np.zeros_like(uv)
    lib().oc_undistort(_p(_f32(K), C.c_float), _p(_f32(dist), C.c_float), C.c_int64(uv.size // 2),
                       _p(uv, C.c_float), _p(out, C.c_float))
    return out


def estimate_homography(p1, p2):
    p1, p2 = _f32(p1), _f32(p2)
    H = np.zeros(9, dtype=np.float32)
    lib().oc_estimate_homography(C.c_int64(p1.shape[0]), _p(p1, C.c_float), C.c_int32(p1.shape[1]),
                                 _p(p2, C.c_float), C.c_int32(p2.shape[1]), _p(H, C.c_float))
    return H.reshape(3, 3)


def estimate_k_from_homographies(Hs):
    Hs = _f32(Hs).reshape(-1, 9)
    K = np.zeros(9, dtype=np.float32)
    lib().oc_estimate_k_from_homographies(C.c_int64(Hs.shape[0]), _p(Hs, C.c_float), _p(K, C.c_float))
    return K.reshape(3, 3)


def recover_extrinsics(K_inv, H):
    R = np.zeros(9, dtype=np.float32)
    t = np.zeros(3, dtype=np.float32)
    lib().oc_recover_extrinsics(_p(_f32(K_inv), C.c_float), _p(_f32(H), C.c_float),
                                _p(R, C.c_float), _p(t, C.c_float))
    return R.reshape(3, 3), t


def fix_rotation_matrix(R):
    out = np.zeros(9, dtype=np.float32)
    lib().oc_fix_rotation_matrix(_p(_f32(R), C.c_float), _p(out, C.c_float))
    return out.reshape(3, 3)


def zhang_init(offsets, uv, xyz):
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    F = len(offsets) - 1
    uv, xyz = _f32(uv), _f32(xyz)
    K = np.zeros(9, dtype=np.float32)
    q = np.zeros((F, 4), dtype=np.float32)
    t = np.zeros((F, 3), dtype=np.float32)
    lib().oc_zhang_init(C.c_int64(F), _p(offsets, C.c_int64), _p(uv, C.c_float), _p(xyz, C.c_float),
                        _p(K, C.c_float), _p(q, C.c_float), _p(t, C.c_float))
    return K.reshape(3, 3), q, t


# Fixture constants of src/test_calibrator.cpp:11-21
FIXTURE_W, FIXTURE_H = 1600, 1000
FIXTURE_K = np.array([[1000, 0, 800], [0, 1000, 500], [0, 0, 1]], dtype=np.float32)
FIXTURE_DIST = np.array([-4.0e-2, 5e-4, 1.0e-3, 2.0e-5, -3e-4], dtype=np.float32)
FIXTURE_NOISE = 0.5


class Generator:
    """DataGenerator restatement (src/data_generator.cpp), mt19937 seed 0."""

    def __init__(self, width=FIXTURE_W, height=FIXTURE_H, K=FIXTURE_K, dist=FIXTURE_DIST,
                 noise=FIXTURE_NOISE):
        self._g = C.c_void_p(lib().oc_generator_create(C.c_int32(width), C.c_int32(height)))
        lib().oc_generator_set_k(self._g, _p(_f32(K), C.c_float))
        lib().oc_generator_set_distortion(self._g, _p(_f32(dist), C.c_float))
        lib().oc_generator_set_noise(self._g, C.c_float(noise))

    def __del__(self):
        if getattr(self, "_g", None):
            lib().oc_generator_destroy(self._g)
            self._g = None

    def planar(self, num_p=100):
        uv = np.zeros((num_p, 2), dtype=np.float32)
        xyz = np.zeros((num_p, 3), dtype=np.float32)
        lib().oc_generator_planar(self._g, C.c_int32(num_p), _p(uv, C.c_float), _p(xyz, C.c_float))
        return uv, xyz

    def points(self, num_p=100):
        uv = np.zeros((num_p, 2), dtype=np.float32)
        xyz = np.zeros((num_p, 3), dtype=np.float32)
        lib().oc_generator_points(self._g, C.c_int32(num_p), _p(uv, C.c_float), _p(xyz, C.c_float))
        return uv, xyz


def make_intrinsics_problem(n_frames, pts_per_frame, **gen_kw):
    """Synthetic single-camera problem: one GetDistortedPointsPlanar call per frame
    (test_calibrator.cpp:52-60). pts_per_frame may be an int or a per-frame sequence (ragged)."""
    g = Generator(**gen_kw)
    counts = [pts_per_frame] * n_frames if np.isscalar(pts_per_frame) else list(pts_per_frame)
    uvs, xyzs = [], []
    for m in counts:
        uv, xyz = g.planar(int(m))
        uvs.append(uv)
        xyzs.append(xyz)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return offsets, np.concatenate(uvs), np.concatenate(xyzs)


def rig_scenario(n_cams, n_frames, pts_per_frame, seed=0):
    C_, F, M = n_cams, n_frames, pts_per_frame
    cam_T = np.zeros((C_, 16), dtype=np.float32)
    cam_T_true = np.zeros((C_, 16), dtype=np.float32)
    frame_T = np.zeros((F, 16), dtype=np.float32)
    world = np.zeros((F * M, 3), dtype=np.float32)
    n = F * M * C_
    obs_cam = np.zeros(n, dtype=np.uint32)
    obs_world = np.zeros(n, dtype=np.uint64)
    obs_uv = np.zeros((n, 2), dtype=np.float32)
    lib().oc_rig_scenario(C.c_int32(C_), C.c_int32(F), C.c_int32(M), C.c_uint32(seed),
                          _p(cam_T, C.c_float), _p(cam_T_true, C.c_float), _p(frame_T, C.c_float),
                          _p(world, C.c_float), _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64),
                          _p(obs_uv, C.c_float))
    offsets = (np.arange(F + 1) * M * C_).astype(np.int64)
    return dict(cam_T=cam_T, cam_T_true=cam_T_true, frame_T=frame_T, world_xyz=world,
                obs_cam=obs_cam, obs_world=obs_world, obs_uv=obs_uv, frame_offsets=offsets,
                cam_frozen=np.array([1] + [0] * (C_ - 1), dtype=np.uint8))


def affine_to_qt(T16):
    T16 = _f32(T16).reshape(-1, 16)
    q = np.zeros((T16.shape[0], 4))
    t = np.zeros((T16.shape[0], 3))
    for i in range(T16.shape[0]):
        lib().oc_affine_to_qt(_p(T16[i], C.c_float), _p(q[i], C.c_double), _p(t[i], C.c_double))
    return q, t


def qt_to_affine(q, t):
    q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
    T = np.zeros((q.shape[0], 16), dtype=np.float32)
    for i in range(q.shape[0]):
        lib().oc_qt_to_affine(_p(q[i], C.c_double), _p(t[i], C.c_double), _p(T[i], C.c_float))
    return T
