"""ctypes binding of the C ABI declared in include/cc_solver.h (libcc_hip.so).

This is the boundary the parity tests and bench.py call through. There is no CPU fallback: if the
shared library is missing, or no gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CC_LIB_PATH", os.path.join(_HERE, "libcc_hip.so"))

K_NAMES = ["sweep", "decide", "elim", "solve", "allreduce", "update", "reduce"]
TERMINATION = {0: "NO_CONVERGENCE", 1: "GRADIENT", 2: "PARAMETER", 3: "FUNCTION",
               4: "FAILURE_INVALID_STEPS", 5: "MIN_RADIUS", 6: "FAILURE_EXCHANGE"}


class CcError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32),
        ("use_nonmonotonic_steps", C.c_int32),
        ("max_consecutive_nonmonotonic_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("max_consecutive_invalid_steps", C.c_int32),
        ("check_interval", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("use_graph", C.c_int32),
        ("profile_kernels", C.c_int32),
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
        ("sweeps", C.c_int32),
        ("kernel_ms", C.c_double * 8),
        ("kernel_launches", C.c_int32 * 8),
        ("kernel_idle_ms", C.c_double * 8),
        ("kernel_idle_launches", C.c_int32 * 8),
    ]


# every symbol include/cc_solver.h declares
EXPORTED_SYMBOLS = [
    "cc_options_init", "cc_last_error", "cc_version", "cc_device_count", "cc_release_caches", "cc_parallel_for", "cc_parallel_parts", "cc_host_pool_threads", "cc_last_call_solver_status",
    "cc_intrinsics_create", "cc_intrinsics_destroy", "cc_intrinsics_set_state",
    "cc_intrinsics_reset", "cc_intrinsics_get_state", "cc_intrinsics_eval",
    "cc_intrinsics_solve", "cc_intrinsics_solver_form", "cc_intrinsics_solver_status", "cc_intrinsics_profile_sweep", "cc_intrinsics_profile_solve", "cc_intrinsics_optimize", "cc_intrinsics_estimate", "cc_intrinsics_estimate_views", "cc_intrinsics_optimize_views", "cc_host_staging_acquire", "cc_host_staging_release", "cc_last_call_timing", "cc_comm_get_unique_id",
    "cc_intrinsics_comm_init", "cc_intrinsics_exchange_export", "cc_intrinsics_exchange_attach", "cc_partition_frames", "cc_distort", "cc_undistort",
    "cc_rig_create", "cc_rig_destroy", "cc_rig_set_state", "cc_rig_reset", "cc_rig_solve",
    "cc_rig_get_state", "cc_rig_solver_form", "cc_rig_solver_status", "cc_rig_eval", "cc_rig_optimize", "cc_rig_comm_init", "cc_rig_exchange_export", "cc_rig_exchange_attach", "cc_rigk_create",
    "cc_rigk_set_intrinsics", "cc_rigk_get_intrinsics", "cc_rigk_create_per_camera", "cc_rigk_set_camera_intrinsics",
    "cc_rigk_get_camera_intrinsics", "cc_zhang_init", "cc_intrinsics_optimize_multi", "cc_rig_optimize_multi", "cc_rig_optimize_frames", "cc_rig_optimize_columns",
]
# every symbol include/cc_harness.h declares (synthetic-input harness, host code)
HARNESS_SYMBOLS = [
    "cc_generator_create", "cc_generator_destroy", "cc_generator_set_k", "cc_generator_set_distortion",
    "cc_generator_set_noise", "cc_generator_planar", "cc_generator_points", "cc_rig_scenario", "cc_affine_to_qt",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CcError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.cc_last_error.restype = C.c_char_p
        _lib.cc_version.restype = C.c_char_p
        _lib.cc_generator_create.restype = C.c_void_p
        _lib.cc_generator_planar.restype = C.c_int64
        _lib.cc_generator_points.restype = C.c_int64
    return _lib


def _check(rc):
    if rc != 0:
        raise CcError(f"cc error {rc}: {lib().cc_last_error().decode()}")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_options(**kw):
    o = Options()
    lib().cc_options_init(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def device_count():
    return int(lib().cc_device_count())


def partition_frames(frame_offsets, nranks):
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    first = np.zeros(nranks + 1, dtype=np.int64)
    _check(lib().cc_partition_frames(C.c_int64(len(off) - 1), _p(off, C.c_int64), C.c_int32(nranks),
                                     _p(first, C.c_int64)))
    return first


def _summary_dict(s, log):
    names = [f[0] for f in Iteration._fields_]
    return {
        "iterations": s.iterations, "successful_steps": s.successful_steps,
        "termination": TERMINATION.get(s.termination, str(s.termination)),
        "initial_cost": s.initial_cost, "final_cost": s.final_cost, "seconds": s.seconds,
        "sweeps": s.sweeps,
        "kernel_ms": {K_NAMES[i]: s.kernel_ms[i] for i in range(len(K_NAMES))},
        "kernel_launches": {K_NAMES[i]: s.kernel_launches[i] for i in range(len(K_NAMES))},
        "kernel_idle_ms": {K_NAMES[i]: s.kernel_idle_ms[i] for i in range(len(K_NAMES))},
        "kernel_idle_launches": {K_NAMES[i]: s.kernel_idle_launches[i] for i in range(len(K_NAMES))},
        "log": [{k: getattr(log[i], k) for k in names} for i in range(s.log_len)],
    }


class IntrinsicsProblem:
    """Handle on a single-camera intrinsics problem resident in HBM (cc_intrinsics_*)."""

    def __init__(self, frame_offsets, uv, xyz, device=0):
        self._h = C.c_void_p()
        off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
        self.n_frames = len(off) - 1
        self.n_obs = int(off[-1])
        uv, xyz = _f32(uv), _f32(xyz)
        assert uv.size == 2 * self.n_obs and xyz.size == 3 * self.n_obs
        _check(lib().cc_intrinsics_create(C.c_int32(device), C.c_int64(self.n_frames),
                                          _p(off, C.c_int64), _p(uv, C.c_float), _p(xyz, C.c_float),
                                          C.byref(self._h)))

    def close(self):
        if self._h:
            lib().cc_intrinsics_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, intr, q, t, const_mask=0):
        intr, q, t = _f64(intr), _f64(q), _f64(t)
        assert intr.size == 9 and q.size == 4 * self.n_frames and t.size == 3 * self.n_frames
        _check(lib().cc_intrinsics_set_state(self._h, _p(intr, C.c_double), C.c_uint32(const_mask),
                                             _p(q, C.c_double), _p(t, C.c_double)))

    def reset(self):
        _check(lib().cc_intrinsics_reset(self._h))

    def get_state(self):
        intr = np.zeros(9)
        q = np.zeros((self.n_frames, 4))
        t = np.zeros((self.n_frames, 3))
        _check(lib().cc_intrinsics_get_state(self._h, _p(intr, C.c_double), _p(q, C.c_double),
                                             _p(t, C.c_double)))
        return intr, q, t

    def eval(self, want_blocks=True):
        blocks = np.zeros((self.n_frames, 16, 16)) if want_blocks else None
        cost = C.c_double()
        _check(lib().cc_intrinsics_eval(self._h, _p(blocks, C.c_double) if want_blocks else None,
                                        C.byref(cost)))
        return cost.value, blocks

    def solve(self, options=None, log_capacity=1024):
        opt = options if options is not None else default_options()
        log = (Iteration * max(1, log_capacity))()
        s = Summary()
        s.log = C.cast(log, C.POINTER(Iteration))
        s.log_capacity = log_capacity
        _check(lib().cc_intrinsics_solve(self._h, C.byref(opt), C.byref(s)))
        return _summary_dict(s, log)

    def solve_lean(self, options):
        """cc_intrinsics_solve without a log and without building the Python summary (timing loops: the dictionary
        costs more host time per solve than the C call's own overhead). Returns the iteration count."""
        s = getattr(self, "_lean_summary", None)
        if s is None:
            s = self._lean_summary = Summary()
            s.log = None
            s.log_capacity = 0
        _check(lib().cc_intrinsics_solve(self._h, C.byref(options), C.byref(s)))
        return s.iterations

    def profile_sweep(self, n=50):
        ms = C.c_double()
        _check(lib().cc_intrinsics_profile_sweep(self._h, C.c_int32(n), C.byref(ms)))
        return ms.value

    def solver_form(self):
        """0: two kernels per LM iteration; 1, 2, 4: the persistent per-solve kernel with that many frames per workgroup."""
        return int(lib().cc_intrinsics_solver_form(self._h))

    def solver_status(self):
        """(form, reruns, note): cc_intrinsics_solver_status -- persistent solves that gave up and were run again, and why."""
        form, reruns = C.c_int32(), C.c_int32()
        note = C.create_string_buffer(2048)
        _check(lib().cc_intrinsics_solver_status(self._h, C.byref(form), C.byref(reruns), note, 2048))
        return form.value, reruns.value, note.value.decode()

    def profile_solve(self, options=None, n=20):
        """Persistent form: (average ms per launch = per complete solve, evaluations per launch), hipEvents on the solver's stream."""
        options = options if options is not None else default_options()
        ms, sw = C.c_double(), C.c_int32()
        _check(lib().cc_intrinsics_profile_solve(self._h, C.byref(options), C.c_int32(n), C.byref(ms), C.byref(sw)))
        return ms.value, sw.value

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().cc_intrinsics_comm_init(self._h, buf, C.c_int32(rank), C.c_int32(nranks)))

    def exchange_export(self):
        """Allocate this rank's mailbox; returns the 64-byte IPC handle to all-gather."""
        buf = (C.c_uint8 * 64)()
        _check(lib().cc_intrinsics_exchange_export(self._h, buf))
        return bytes(buf)

    def exchange_attach(self, rank, handles):
        """handles: every rank's 64-byte handle, in rank order."""
        blob = b"".join(bytes(x) for x in handles)
        assert len(blob) == 64 * len(handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(lib().cc_intrinsics_exchange_attach(self._h, C.c_int32(rank), C.c_int32(len(handles)), buf))


def comm_get_unique_id():
    buf = (C.c_uint8 * 128)()
    _check(lib().cc_comm_get_unique_id(buf))
    return bytes(buf)


def _views(off, uv, xyz):
    """(uv_views, xyz_views, counts, keep-alive) for the *_views entry points: one pointer per view, each view its own
    array (as a vector<Points2D> holds them), not slices of one block."""
    F = len(off) - 1
    parts_uv = [np.array(uv.reshape(-1, 2)[off[f]:off[f + 1]], dtype=np.float32, order="C") for f in range(F)]
    parts_xyz = [np.array(xyz.reshape(-1, 3)[off[f]:off[f + 1]], dtype=np.float32, order="C") for f in range(F)]
    fp = C.POINTER(C.c_float)
    uv_views = (fp * F)(*[a.ctypes.data_as(fp) for a in parts_uv])
    xyz_views = (fp * F)(*[a.ctypes.data_as(fp) for a in parts_xyz])
    counts = np.ascontiguousarray(np.diff(off), dtype=np.int64)
    return uv_views, xyz_views, counts, (parts_uv, parts_xyz)


def intrinsics_optimize(frame_offsets, uv, xyz, intr, q, t, const_mask=0, options=None, device=0,
                        log_capacity=1024, devices=None, views=False):
    """One-shot cc_intrinsics_optimize (devices=[...]: cc_intrinsics_optimize_multi, one host thread driving several
    devices; views=True: cc_intrinsics_optimize_views, every view handed over as its own array). Returns (intr, q, t, summary)."""
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(off) - 1
    uv, xyz = _f32(uv), _f32(xyz)
    intr, q, t = _f64(intr).copy(), _f64(q).copy(), _f64(t).copy()
    opt = options if options is not None else default_options()
    log = (Iteration * max(1, log_capacity))()
    s = Summary()
    s.log = C.cast(log, C.POINTER(Iteration))
    s.log_capacity = log_capacity
    if devices is not None:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        _check(lib().cc_intrinsics_optimize_multi(C.byref(opt), C.c_int32(len(devs)), _p(devs, C.c_int32), C.c_int64(F),
                                                  _p(off, C.c_int64), _p(uv, C.c_float), _p(xyz, C.c_float),
                                                  _p(intr, C.c_double), C.c_uint32(const_mask),
                                                  _p(q, C.c_double), _p(t, C.c_double), C.byref(s)))
        return intr, q, t, _summary_dict(s, log)
    if views:
        uv_views, xyz_views, counts, _keep = _views(off, uv, xyz)
        _check(lib().cc_intrinsics_optimize_views(C.byref(opt), C.c_int32(device), C.c_int64(F), uv_views, xyz_views,
                                                  _p(counts, C.c_int64), _p(intr, C.c_double), C.c_uint32(const_mask),
                                                  _p(q, C.c_double), _p(t, C.c_double), C.byref(s)))
        return intr, q, t, _summary_dict(s, log)
    _check(lib().cc_intrinsics_optimize(C.byref(opt), C.c_int32(device), C.c_int64(F),
                                        _p(off, C.c_int64), _p(uv, C.c_float), _p(xyz, C.c_float),
                                        _p(intr, C.c_double), C.c_uint32(const_mask),
                                        _p(q, C.c_double), _p(t, C.c_double), C.byref(s)))
    return intr, q, t, _summary_dict(s, log)


def intrinsics_estimate(frame_offsets, uv, xyz, distortion5=None, const_mask=0, options=None, device=0, log_capacity=1024,
                        views=False):
    """cc_intrinsics_estimate (= Calibrator::Estimate: Zhang initialisation + solve, one upload; views=True:
    cc_intrinsics_estimate_views, every view handed over as its own array and packed by the library under its upload).
    Returns (K_init float32 3x3, intr, q, t, summary)."""
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(off) - 1
    uv, xyz = _f32(uv), _f32(xyz)
    opt = options if options is not None else default_options()
    log = (Iteration * max(1, log_capacity))()
    s = Summary()
    s.log = C.cast(log, C.POINTER(Iteration))
    s.log_capacity = log_capacity
    K = np.zeros(9, dtype=np.float32)
    intr, q, t = np.zeros(9), np.zeros((F, 4)), np.zeros((F, 3))
    d5 = _f64(distortion5) if distortion5 is not None else None
    if views:
        uv_views, xyz_views, counts, _keep = _views(off, uv, xyz)
        _check(lib().cc_intrinsics_estimate_views(C.byref(opt), C.c_int32(device), C.c_int64(F), uv_views, xyz_views,
                                                  _p(counts, C.c_int64), _p(d5, C.c_double) if d5 is not None else None,
                                                  C.c_uint32(const_mask), _p(K, C.c_float), _p(intr, C.c_double),
                                                  _p(q, C.c_double), _p(t, C.c_double), C.byref(s)))
        return K.reshape(3, 3), intr, q, t, _summary_dict(s, log)
    _check(lib().cc_intrinsics_estimate(C.byref(opt), C.c_int32(device), C.c_int64(F), _p(off, C.c_int64), _p(uv, C.c_float),
                                        _p(xyz, C.c_float), _p(d5, C.c_double) if d5 is not None else None,
                                        C.c_uint32(const_mask), _p(K, C.c_float), _p(intr, C.c_double), _p(q, C.c_double),
                                        _p(t, C.c_double), C.byref(s)))
    return K.reshape(3, 3), intr, q, t, _summary_dict(s, log)


HUBER_A = float(np.float32(3.0) / np.float32(500.0))  # extrinsics_calibrator.cpp:176


def release_caches():
    """cc_release_caches: hand the library's cached device / pinned memory back (idle pieces only)."""
    lib().cc_release_caches.restype = None
    lib().cc_release_caches()


class ObsLayout(C.Structure):
    _fields_ = [("stride", C.c_int64), ("camera_offset", C.c_int64), ("world_offset", C.c_int64), ("uv_offset", C.c_int64),
                ("cost_offset", C.c_int64)]


# one sighting as ExtrinsicsCalibrator keeps it (camera, point in frame, global point, normalised image point, cost)
SIGHTING = np.dtype([("camera", "<u8"), ("point_in_frame", "<u8"), ("world", "<u8"), ("uv", "<f4", (2,)), ("cost", "<f8")])


def rig_optimize_frames(n_cams, frame_offsets, obs_cam, obs_world, obs_uv, world_xyz, cam_q, cam_t, cam_frozen,
                        frame_q, frame_t, huber_a=HUBER_A, options=None, device=0, log_capacity=2048):
    """cc_rig_optimize_frames: the observations handed over frame by frame as arrays of SIGHTING records (one numpy array per
    frame, as the C++ class holds one vector per frame); the costs come back inside the records.
    Returns (cam_q, cam_t, frame_q, frame_t, obs_cost, summary) like rig_optimize."""
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(off) - 1
    obs_cam, obs_world = np.asarray(obs_cam), np.asarray(obs_world)
    uv = _f32(obs_uv).reshape(-1, 2)
    world_xyz = _f32(world_xyz)
    frames = []
    for f in range(F):
        a = np.zeros(int(off[f + 1] - off[f]), dtype=SIGHTING)
        a["camera"] = obs_cam[off[f]:off[f + 1]]
        a["world"] = obs_world[off[f]:off[f + 1]]
        a["uv"] = uv[off[f]:off[f + 1]]
        a["cost"] = -1.0
        frames.append(a)
    ptrs = (C.c_void_p * F)(*[a.ctypes.data if len(a) else None for a in frames])
    counts = np.ascontiguousarray(np.diff(off), dtype=np.int64)
    lay = ObsLayout(SIGHTING.itemsize, SIGHTING.fields["camera"][1], SIGHTING.fields["world"][1], SIGHTING.fields["uv"][1],
                    SIGHTING.fields["cost"][1])
    frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
    cam_q, cam_t = _f64(cam_q).copy(), _f64(cam_t).copy()
    frame_q, frame_t = _f64(frame_q).copy(), _f64(frame_t).copy()
    opt = options if options is not None else default_options(max_iterations=1000)
    log = (Iteration * max(1, log_capacity))()
    s = Summary()
    s.log = C.cast(log, C.POINTER(Iteration))
    s.log_capacity = log_capacity
    _check(lib().cc_rig_optimize_frames(C.byref(opt), C.c_int32(device), C.c_int64(n_cams), C.c_int64(F),
                                        C.c_int64(world_xyz.size // 3), ptrs, _p(counts, C.c_int64), C.byref(lay),
                                        _p(world_xyz, C.c_float), _p(cam_q, C.c_double), _p(cam_t, C.c_double),
                                        _p(frozen, C.c_uint8), _p(frame_q, C.c_double), _p(frame_t, C.c_double),
                                        C.c_double(huber_a), C.byref(s)))
    cost = np.concatenate([a["cost"] for a in frames]) if F else np.zeros(0)
    return cam_q, cam_t, frame_q, frame_t, cost, _summary_dict(s, log)


class ObsColumns(C.Structure):
    _fields_ = [("camera", C.POINTER(C.c_void_p)), ("camera_stride", C.c_int64), ("camera_width", C.c_int32),
                ("world", C.POINTER(C.c_void_p)), ("world_stride", C.c_int64), ("world_width", C.c_int32),
                ("uv", C.POINTER(C.c_void_p)), ("uv_stride", C.c_int64),
                ("cost", C.POINTER(C.c_void_p)), ("cost_stride", C.c_int64)]


def rig_optimize_columns(n_cams, frame_offsets, obs_cam, obs_world, obs_uv, world_xyz, cam_q, cam_t, cam_frozen,
                         frame_q, frame_t, huber_a=HUBER_A, options=None, device=0, log_capacity=2048,
                         camera_dtype=np.uint32, world_dtype=np.uint64, want_cost=True):
    """cc_rig_optimize_columns: the observations handed over frame by frame as four arrays per frame (camera ids, world point ids,
    image points, costs -- what the C++ class holds since round 5); ids 4 or 8 bytes wide.
    Returns (cam_q, cam_t, frame_q, frame_t, obs_cost, summary) like rig_optimize (obs_cost None with want_cost=False)."""
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(off) - 1
    obs_cam, obs_world = np.asarray(obs_cam), np.asarray(obs_world)
    uv = _f32(obs_uv).reshape(-1, 2)
    world_xyz = _f32(world_xyz)
    cols = {"camera": [], "world": [], "uv": [], "cost": []}
    for f in range(F):
        a, b = int(off[f]), int(off[f + 1])
        cols["camera"].append(np.ascontiguousarray(obs_cam[a:b], dtype=camera_dtype))
        cols["world"].append(np.ascontiguousarray(obs_world[a:b], dtype=world_dtype))
        cols["uv"].append(np.ascontiguousarray(uv[a:b]))
        cols["cost"].append(np.full(b - a, -1.0))
    ptr = {k: (C.c_void_p * F)(*[x.ctypes.data if len(x) else None for x in v]) for k, v in cols.items()}
    oc = ObsColumns(ptr["camera"], np.dtype(camera_dtype).itemsize, np.dtype(camera_dtype).itemsize,
                    ptr["world"], np.dtype(world_dtype).itemsize, np.dtype(world_dtype).itemsize,
                    ptr["uv"], 8, ptr["cost"] if want_cost else None, 8)
    counts = np.ascontiguousarray(np.diff(off), dtype=np.int64)
    frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
    cam_q, cam_t = _f64(cam_q).copy(), _f64(cam_t).copy()
    frame_q, frame_t = _f64(frame_q).copy(), _f64(frame_t).copy()
    opt = options if options is not None else default_options(max_iterations=1000)
    log = (Iteration * max(1, log_capacity))()
    s = Summary()
    s.log = C.cast(log, C.POINTER(Iteration))
    s.log_capacity = log_capacity
    _check(lib().cc_rig_optimize_columns(C.byref(opt), C.c_int32(device), C.c_int64(n_cams), C.c_int64(F),
                                         C.c_int64(world_xyz.size // 3), C.byref(oc), _p(counts, C.c_int64),
                                         _p(world_xyz, C.c_float), _p(cam_q, C.c_double), _p(cam_t, C.c_double),
                                         _p(frozen, C.c_uint8), _p(frame_q, C.c_double), _p(frame_t, C.c_double),
                                         C.c_double(huber_a), C.byref(s)))
    cost = (np.concatenate(cols["cost"]) if F else np.zeros(0)) if want_cost else None
    return cam_q, cam_t, frame_q, frame_t, cost, _summary_dict(s, log)


class RigProblem:
    """Handle on a rig pose problem resident in HBM (cc_rig_*)."""

    def __init__(self, n_cams, frame_offsets, obs_cam, obs_world, obs_uv, world_xyz, cam_frozen,
                 huber_a=HUBER_A, device=0, with_intrinsics=False):
        """with_intrinsics=True: the extension cc_rigk_create (pixel observations, 9 shared intrinsics);
        with_intrinsics="per_camera": cc_rigk_create_per_camera (one set of 9 per camera)."""
        self._h = C.c_void_p()
        self.with_intrinsics = bool(with_intrinsics)
        self.per_camera = with_intrinsics == "per_camera"
        off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
        self.n_cams, self.n_frames, self.n_obs = int(n_cams), len(off) - 1, int(off[-1])
        obs_cam = np.ascontiguousarray(obs_cam, dtype=np.uint32)
        obs_world = np.ascontiguousarray(obs_world, dtype=np.uint64)
        obs_uv, world_xyz = _f32(obs_uv), _f32(world_xyz)
        frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
        create = (lib().cc_rigk_create_per_camera if self.per_camera else lib().cc_rigk_create) if with_intrinsics else lib().cc_rig_create
        _check(create(C.c_int32(device), C.c_int64(n_cams), C.c_int64(self.n_frames),
                                   C.c_int64(world_xyz.size // 3), _p(off, C.c_int64),
                                   _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64),
                                   _p(obs_uv, C.c_float), _p(world_xyz, C.c_float),
                                   _p(frozen, C.c_uint8), C.c_double(huber_a), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().cc_rig_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, cam_q, cam_t, frame_q, frame_t):
        cam_q, cam_t, frame_q, frame_t = _f64(cam_q), _f64(cam_t), _f64(frame_q), _f64(frame_t)
        assert cam_q.size == 4 * self.n_cams and frame_q.size == 4 * self.n_frames
        _check(lib().cc_rig_set_state(self._h, _p(cam_q, C.c_double), _p(cam_t, C.c_double),
                                      _p(frame_q, C.c_double), _p(frame_t, C.c_double)))

    def set_intrinsics(self, intr9, const_mask=0):
        intr9 = _f64(intr9)
        assert intr9.size == 9
        _check(lib().cc_rigk_set_intrinsics(self._h, _p(intr9, C.c_double), C.c_uint32(const_mask)))

    def get_intrinsics(self):
        out = np.zeros(9)
        _check(lib().cc_rigk_get_intrinsics(self._h, _p(out, C.c_double)))
        return out

    def set_camera_intrinsics(self, camera, intr9, const_mask=0):
        intr9 = _f64(intr9)
        assert intr9.size == 9
        _check(lib().cc_rigk_set_camera_intrinsics(self._h, C.c_int64(camera), _p(intr9, C.c_double), C.c_uint32(const_mask)))

    def get_camera_intrinsics(self, camera=None):
        """One camera's set, or (camera=None) an [n_cams, 9] array of all of them."""
        if camera is None:
            return np.stack([self.get_camera_intrinsics(c) for c in range(self.n_cams)])
        out = np.zeros(9)
        _check(lib().cc_rigk_get_camera_intrinsics(self._h, C.c_int64(camera), _p(out, C.c_double)))
        return out

    def reset(self):
        _check(lib().cc_rig_reset(self._h))

    def solve(self, options=None, log_capacity=2048):
        opt = options if options is not None else default_options(max_iterations=1000)
        log = (Iteration * max(1, log_capacity))()
        s = Summary()
        s.log = C.cast(log, C.POINTER(Iteration))
        s.log_capacity = log_capacity
        _check(lib().cc_rig_solve(self._h, C.byref(opt), C.byref(s)))
        return _summary_dict(s, log)

    def get_state(self, want_cost=True):
        cq, ct = np.zeros((self.n_cams, 4)), np.zeros((self.n_cams, 3))
        fq, ft = np.zeros((self.n_frames, 4)), np.zeros((self.n_frames, 3))
        cost = np.zeros(self.n_obs) if want_cost else None
        _check(lib().cc_rig_get_state(self._h, _p(cq, C.c_double), _p(ct, C.c_double), _p(fq, C.c_double),
                                      _p(ft, C.c_double), _p(cost, C.c_double) if want_cost else None))
        return cq, ct, fq, ft, cost

    def solver_form(self):
        """2: the whole solve as one launch of the lean persistent kernel; 0: three kernels per LM iteration."""
        return int(lib().cc_rig_solver_form(self._h))

    def solver_status(self):
        """(form, reruns, note): cc_rig_solver_status -- lean persistent solves that gave up and were run again, and why."""
        form, reruns = C.c_int32(), C.c_int32()
        note = C.create_string_buffer(1024)
        _check(lib().cc_rig_solver_status(self._h, C.byref(form), C.byref(reruns), note, 1024))
        return form.value, reruns.value, note.value.decode()

    def eval(self):
        c = C.c_double()
        _check(lib().cc_rig_eval(self._h, C.byref(c)))
        return c.value

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().cc_rig_comm_init(self._h, buf, C.c_int32(rank), C.c_int32(nranks)))

    def exchange_export(self):
        buf = (C.c_uint8 * 64)()
        _check(lib().cc_rig_exchange_export(self._h, buf))
        return bytes(buf)

    def exchange_attach(self, rank, handles):
        """Collective: every rank calls it with all handles in rank order."""
        blob = b"".join(bytes(x) for x in handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(lib().cc_rig_exchange_attach(self._h, C.c_int32(rank), C.c_int32(len(handles)), buf))


def rig_optimize(n_cams, frame_offsets, obs_cam, obs_world, obs_uv, world_xyz, cam_q, cam_t, cam_frozen,
                 frame_q, frame_t, huber_a=HUBER_A, options=None, device=0, log_capacity=2048, devices=None):
    """One-shot cc_rig_optimize (devices=[...]: cc_rig_optimize_multi). Returns (cam_q, cam_t, frame_q, frame_t, obs_cost, summary)."""
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(off) - 1
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.uint32)
    obs_world = np.ascontiguousarray(obs_world, dtype=np.uint64)
    obs_uv, world_xyz = _f32(obs_uv), _f32(world_xyz)
    frozen = np.ascontiguousarray(cam_frozen, dtype=np.uint8)
    cam_q, cam_t = _f64(cam_q).copy(), _f64(cam_t).copy()
    frame_q, frame_t = _f64(frame_q).copy(), _f64(frame_t).copy()
    cost = np.zeros(len(obs_cam))
    opt = options if options is not None else default_options(max_iterations=1000)
    log = (Iteration * max(1, log_capacity))()
    s = Summary()
    s.log = C.cast(log, C.POINTER(Iteration))
    s.log_capacity = log_capacity
    if devices is not None:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        _check(lib().cc_rig_optimize_multi(C.byref(opt), C.c_int32(len(devs)), _p(devs, C.c_int32), C.c_int64(n_cams), C.c_int64(F),
                                           C.c_int64(world_xyz.size // 3), _p(off, C.c_int64), _p(obs_cam, C.c_uint32),
                                           _p(obs_world, C.c_uint64), _p(obs_uv, C.c_float), _p(world_xyz, C.c_float),
                                           _p(cam_q, C.c_double), _p(cam_t, C.c_double), _p(frozen, C.c_uint8),
                                           _p(frame_q, C.c_double), _p(frame_t, C.c_double), C.c_double(huber_a),
                                           _p(cost, C.c_double), C.byref(s)))
        return cam_q, cam_t, frame_q, frame_t, cost, _summary_dict(s, log)
    _check(lib().cc_rig_optimize(C.byref(opt), C.c_int32(device), C.c_int64(n_cams), C.c_int64(F),
                                 C.c_int64(world_xyz.size // 3), _p(off, C.c_int64), _p(obs_cam, C.c_uint32),
                                 _p(obs_world, C.c_uint64), _p(obs_uv, C.c_float), _p(world_xyz, C.c_float),
                                 _p(cam_q, C.c_double), _p(cam_t, C.c_double), _p(frozen, C.c_uint8),
                                 _p(frame_q, C.c_double), _p(frame_t, C.c_double), C.c_double(huber_a),
                                 _p(cost, C.c_double), C.byref(s)))
    return cam_q, cam_t, frame_q, frame_t, cost, _summary_dict(s, log)


def zhang_init(frame_offsets, uv, xyz, device=0, want_homographies=False):
    """cc_zhang_init: returns (K 3x3 float32, q [F,4] float32, t [F,3] float32[, H [F,3,3]])."""
    off = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    F = len(off) - 1
    uv, xyz = _f32(uv), _f32(xyz)
    K = np.zeros(9, dtype=np.float32)
    q = np.zeros((F, 4), dtype=np.float32)
    t = np.zeros((F, 3), dtype=np.float32)
    H = np.zeros((F, 3, 3), dtype=np.float32)
    _check(lib().cc_zhang_init(C.c_int32(device), C.c_int64(F), _p(off, C.c_int64), _p(uv, C.c_float),
                               _p(xyz, C.c_float), _p(K, C.c_float), _p(q, C.c_float), _p(t, C.c_float),
                               _p(H, C.c_float) if want_homographies else None))
    return (K.reshape(3, 3), q, t, H) if want_homographies else (K.reshape(3, 3), q, t)


def distort(K, dist, xy, device=0):
    xy = _f32(xy)
    out = np.zeros_like(xy)
    _check(lib().cc_distort(C.c_int32(device), _p(_f32(K), C.c_float), _p(_f32(dist), C.c_float),
                            C.c_int64(xy.size // 2), _p(xy, C.c_float), _p(out, C.c_float)))
    return out


def undistort(K, dist, uv, device=0):
    uv = _f32(uv)
    out = np.zeros_like(uv)
    _check(lib().cc_undistort(C.c_int32(device), _p(_f32(K), C.c_float), _p(_f32(dist), C.c_float),
                              C.c_int64(uv.size // 2), _p(uv, C.c_float), _p(out, C.c_float)))
    return out


# ---- synthetic-input harness (include/cc_harness.h): the reference's DataGenerator without OpenCV ----
# fixture constants of src/test_calibrator.cpp:11-21
FIXTURE_W, FIXTURE_H = 1600, 1000
FIXTURE_K = np.array([[1000, 0, 800], [0, 1000, 500], [0, 0, 1]], dtype=np.float32)
FIXTURE_DIST = np.array([-4.0e-2, 5e-4, 1.0e-3, 2.0e-5, -3e-4], dtype=np.float32)
FIXTURE_NOISE = 0.5


class Generator:
    def __init__(self, width=FIXTURE_W, height=FIXTURE_H, K=FIXTURE_K, dist=FIXTURE_DIST, noise=FIXTURE_NOISE):
        self._g = C.c_void_p(lib().cc_generator_create(C.c_int32(width), C.c_int32(height)))
        lib().cc_generator_set_k(self._g, _p(_f32(K), C.c_float))
        lib().cc_generator_set_distortion(self._g, _p(_f32(dist), C.c_float))
        lib().cc_generator_set_noise(self._g, C.c_float(noise))

    def __del__(self):
        if getattr(self, "_g", None):
            lib().cc_generator_destroy(self._g)
            self._g = None

    def planar(self, num_p=100):
        uv = np.zeros((num_p, 2), dtype=np.float32)
        xyz = np.zeros((num_p, 3), dtype=np.float32)
        n = lib().cc_generator_planar(self._g, C.c_int32(num_p), _p(uv, C.c_float), _p(xyz, C.c_float))
        assert n == num_p
        return uv, xyz

    def points(self, num_p=100):
        uv = np.zeros((num_p, 2), dtype=np.float32)
        xyz = np.zeros((num_p, 3), dtype=np.float32)
        n = lib().cc_generator_points(self._g, C.c_int32(num_p), _p(uv, C.c_float), _p(xyz, C.c_float))
        assert n == num_p
        return uv, xyz


def make_intrinsics_problem(n_frames, pts_per_frame, **gen_kw):
    """One GetDistortedPointsPlanar call per frame (src/test_calibrator.cpp:52-60); pts_per_frame may be ragged."""
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
    """cc_rig_scenario: the reference's rig test scenario (test_extrinsics_calibrator.cpp:48-134) at any size."""
    C_, F, M = int(n_cams), int(n_frames), int(pts_per_frame)
    cam_T = np.zeros((C_, 16), dtype=np.float32)
    cam_T_true = np.zeros((C_, 16), dtype=np.float32)
    frame_T = np.zeros((F, 16), dtype=np.float32)
    world = np.zeros((F * M, 3), dtype=np.float32)
    n = F * M * C_
    obs_cam = np.zeros(n, dtype=np.uint32)
    obs_world = np.zeros(n, dtype=np.uint64)
    obs_uv = np.zeros((n, 2), dtype=np.float32)
    lib().cc_rig_scenario(C.c_int32(C_), C.c_int32(F), C.c_int32(M), C.c_uint32(seed), _p(cam_T, C.c_float),
                          _p(cam_T_true, C.c_float), _p(frame_T, C.c_float), _p(world, C.c_float),
                          _p(obs_cam, C.c_uint32), _p(obs_world, C.c_uint64), _p(obs_uv, C.c_float))
    return dict(cam_T=cam_T, cam_T_true=cam_T_true, frame_T=frame_T, world_xyz=world, obs_cam=obs_cam,
                obs_world=obs_world, obs_uv=obs_uv, frame_offsets=(np.arange(F + 1) * M * C_).astype(np.int64),
                cam_frozen=np.array([1] + [0] * (C_ - 1), dtype=np.uint8))


def affine_to_qt(T16):
    """cc_affine_to_qt on every row: (q [n,4] w x y z, t [n,3]) in fp64."""
    T16 = _f32(T16).reshape(-1, 16)
    q = np.zeros((T16.shape[0], 4))
    t = np.zeros((T16.shape[0], 3))
    for i in range(T16.shape[0]):
        lib().cc_affine_to_qt(_p(T16[i], C.c_float), _p(q[i], C.c_double), _p(t[i], C.c_double))
    return q, t
