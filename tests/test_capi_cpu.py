"""Host-side checks of the C-ABI library that need no GPU: it loads, exports every symbol
include/cc_solver.h declares, its host logic (frame partition, option defaults) is right, and
compute entry points fail loudly when no device exists (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from camera_calibrator_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="cc_solver.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 15
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTED_SYMBOLS) == declared
    harness = _declared_symbols("cc_harness.h")
    assert sorted(capi.HARNESS_SYMBOLS) == harness and all(hasattr(lib, s) for s in harness)
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["cc_harness.h", "cc_solver.h"]


def test_product_generator_is_bit_identical_to_the_oracle_generator():
    """The harness DataGenerator (camera_calibrator_amd/csrc/data_generator.cpp) and the oracle's
    restatement (oracle/oracle.cpp) are two independent codes of src/data_generator.cpp."""
    from oracle import pyoracle as po
    for frames, pts in [(5, 100), (7, [8, 64, 65, 300, 5, 257, 128])]:
        a = capi.make_intrinsics_problem(frames, pts)
        b = po.make_intrinsics_problem(frames, pts)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    g1, g2 = capi.Generator(), po.Generator()
    u1, x1 = g1.points(50)
    u2, x2 = g2.points(50)
    assert np.array_equal(u1, u2) and np.array_equal(x1, x2)
    z = capi.Generator(noise=0.0).planar(20)
    assert np.all(z[1][:, 2] == 0)


def test_product_rig_scenario_is_bit_identical_to_the_oracle_scenario():
    """cc_rig_scenario (camera_calibrator_amd/csrc/rig_scenario.cpp) and the oracle's oc_rig_scenario are two
    independent codes of test_extrinsics_calibrator.cpp:9-134; likewise the Affine3f -> (q, t) conversion."""
    from oracle import pyoracle as po
    for cams, frames, pts in [(2, 30, 4), (5, 9, 33)]:
        a, b = capi.rig_scenario(cams, frames, pts), po.rig_scenario(cams, frames, pts)
        assert sorted(a) == sorted(b) and all(np.array_equal(a[k], b[k]) for k in a)
        for key in ("cam_T", "frame_T"):
            qa, ta = capi.affine_to_qt(a[key])
            qb, tb = po.affine_to_qt(b[key])
            assert np.array_equal(qa, qb) and np.array_equal(ta, tb)
    assert not np.array_equal(capi.rig_scenario(2, 3, 4, seed=1)["obs_uv"], capi.rig_scenario(2, 3, 4)["obs_uv"])


def test_option_defaults_match_reference_call_site():
    o = capi.default_options()
    assert (o.max_iterations, o.use_nonmonotonic_steps) == (100, 1)          # calibrator.cpp:315,319
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert (o.initial_radius, o.max_radius, o.min_radius) == (1e4, 1e16, 1e-32)
    assert (o.min_lm_diagonal, o.max_lm_diagonal, o.min_relative_decrease) == (1e-6, 1e32, 1e-3)


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(capi.Options) == 6 * 4 + 9 * 8 + 2 * 4
    assert C.sizeof(capi.Iteration) == 7 * 8 + 2 * 4
    assert C.sizeof(capi.Summary) == 4 * 4 + 3 * 8 + 8 + 2 * 4 + 2 * (8 * 8 + 8 * 4)


def test_observation_column_structs_are_what_the_c_compiler_lays_out(tmp_path):
    """cc_obs_columns / cc_obs_layout as ctypes sees them against the header compiled by gcc (sizes and every field offset)."""
    import ctypes as C
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("needs gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = {"cc_obs_columns": [f for f, _ in capi.ObsColumns._fields_], "cc_obs_layout": [f for f, _ in capi.ObsLayout._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "cc_solver.h"', "int main(void) {"]
    for st, fs in fields.items():
        src.append('printf("%%zu", sizeof(%s));' % st)
        src += ['printf(" %%zu", offsetof(%s, %s));' % (st, f) for f in fs]
        src.append('printf("\\n");')
    src.append("return 0; }")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    for line, (st, cls) in zip(out, [("cc_obs_columns", capi.ObsColumns), ("cc_obs_layout", capi.ObsLayout)]):
        nums = [int(x) for x in line.split()]
        assert nums[0] == C.sizeof(cls), st
        assert nums[1:] == [getattr(cls, f).offset for f, _ in cls._fields_], st


@pytest.mark.parametrize("nranks", [1, 2, 3, 4, 8])
def test_partition_frames_uniform(nranks):
    off = np.arange(0, 1001) * 500
    first = capi.partition_frames(off, nranks)
    assert first[0] == 0 and first[-1] == 1000 and np.all(np.diff(first) > 0)
    counts = np.diff(off[first])
    assert counts.max() - counts.min() <= 500


def test_partition_frames_ragged_and_degenerate():
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 400, size=137)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for nranks in (2, 5, 8):
        first = capi.partition_frames(off, nranks)
        assert first[0] == 0 and first[-1] == 137 and np.all(np.diff(first) >= 1)
        counts = np.diff(off[first])
        assert counts.max() <= off[-1] / nranks + 400
    # more ranks than frames: trailing ranks get empty shards, nothing is lost
    first = capi.partition_frames([0, 10, 20], 4)
    assert first[0] == 0 and first[-1] == 2 and np.all(np.diff(first) >= 0)


def test_bad_arguments_are_reported():
    with pytest.raises(capi.CcError):
        capi.partition_frames([0, 1], 0)


@pytest.mark.skipif(capi.device_count() > 0, reason="only meaningful without a GPU")
def test_compute_entry_points_fail_loudly_without_gpu():
    with pytest.raises(capi.CcError, match="no HIP device|no CPU fallback"):
        capi.IntrinsicsProblem([0, 4], np.zeros((4, 2)), np.zeros((4, 3)))
    with pytest.raises(capi.CcError):
        capi.distort(np.eye(3), np.zeros(5), np.zeros((3, 2)))


def test_host_worker_pool_runs_every_part_once_and_is_joined_by_release_caches():
    """cc_parallel_for (round 5): the library's process-lifetime worker pool, which the C++ classes use for their fills instead
    of creating threads per call. Every part runs exactly once; two host threads may submit at the same time (one job at a
    time); cc_release_caches joins the workers and the next job starts them again. Host code only: no GPU needed."""
    import ctypes as C
    import threading
    lib = capi.lib()
    lib.cc_parallel_parts.restype = C.c_int32
    lib.cc_parallel_parts.argtypes = [C.c_int64, C.c_int64]
    lib.cc_host_pool_threads.restype = C.c_int32
    FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)
    lib.cc_parallel_for.argtypes = [C.c_int32, FN, C.c_void_p]
    lib.cc_parallel_for.restype = None
    assert lib.cc_parallel_parts(10, 100) == 1 and 1 <= lib.cc_parallel_parts(1 << 24, 1 << 16) <= 16

    def job(parts, counts):
        lock = threading.Lock()

        def body(_ctx, t):
            with lock:
                counts[t] += 1
        cb = FN(body)
        lib.cc_parallel_for(parts, cb, None)

    a, b = [0] * 7, [0] * 12
    ta, tb = threading.Thread(target=job, args=(7, a)), threading.Thread(target=job, args=(12, b))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert a == [1] * 7 and b == [1] * 12
    assert 1 <= lib.cc_host_pool_threads() <= 15
    lib.cc_release_caches()
    assert lib.cc_host_pool_threads() == 0
    c = [0] * 5
    job(5, c)
    assert c == [1] * 5 and lib.cc_host_pool_threads() >= 1
    one = [0]
    job(1, one)            # a single part runs on the calling thread
    assert one == [1]


def test_host_worker_pool_can_be_entered_again_from_one_of_its_parts():
    """the pool's job lock is not recursive -- a part that calls cc_parallel_for (or cc_release_caches) again
    used to deadlock. A nested job now runs its parts inline on the calling part's thread; release from inside a part is a
    no-op for the pool. Run in a child process with a timeout, so that a deadlock is a failure and not a hang."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, threading, ctypes as C
        sys.path.insert(0, %r)
        from camera_calibrator_amd import capi
        lib = capi.lib()
        FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)
        lib.cc_parallel_for.argtypes = [C.c_int32, FN, C.c_void_p]; lib.cc_parallel_for.restype = None
        lib.cc_host_pool_threads.restype = C.c_int32
        lock = threading.Lock()
        inner, outer = [0] * 6, [0] * 5
        def inner_body(_c, t):
            with lock: inner[t] += 1
        icb = FN(inner_body)
        def outer_body(_c, t):
            with lock: outer[t] += 1
            lib.cc_parallel_for(6, icb, None)      # nested: inline on this part's thread
            if t == 2: lib.cc_release_caches()     # from inside a part: the pool's threads stay
        lib.cc_parallel_for(5, FN(outer_body), None)
        assert outer == [1] * 5 and inner == [5] * 6, (outer, inner)
        assert lib.cc_host_pool_threads() >= 1
        lib.cc_parallel_for(5, FN(outer_body), None)   # the pool still works afterwards
        assert outer == [2] * 5 and inner == [10] * 6, (outer, inner)
        print("nested ok")
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "nested ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_last_call_solver_status_is_exported_and_empty_before_any_call():
    import ctypes as C
    lib = capi.lib()
    form, reruns, note = C.c_int32(-1), C.c_int32(-1), C.create_string_buffer(64)
    assert lib.cc_last_call_solver_status(C.byref(form), C.byref(reruns), note, 64) == 0
    assert reruns.value == 0 and note.value == b""
