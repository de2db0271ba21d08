"""Runs the native C++ drop-in program (tests/cpp/test_dropin.cpp, built by __graft_entry__.build()):
the Calibrator / ExtrinsicsCalibrator classes driven from plain C++ through the C ABI, scenarios of the
reference's src/test_calibrator.cpp and src/test_extrinsics_calibrator.cpp."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_native_cpp_program_passes_its_checks():
    exe = os.path.join(HERE, "cpp", "test_dropin")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "all checks passed" in r.stdout


def test_class_surface_lines_of_both_classes():
    """The timing modes bench.py embeds (`class_surface`): one JSON line each, the solves they time converge."""
    import json
    exe = os.path.join(HERE, "cpp", "test_dropin")
    r = subprocess.run([exe, "--class-surface", "40", "60", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["observations"] == 2400 and 0 < line["lm_iterations"] < 100 and line["wall_ms_median"] > 0
    r = subprocess.run([exe, "--class-surface-rig", "3", "60", "20", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["observations"] == 3600
    fresh, again = line["fresh_object"], line["same_object_again"]
    assert 0 < fresh["lm_iterations"] < 1000 and again["lm_iterations"] <= fresh["lm_iterations"]
    assert fresh["wall_ms_median"] >= fresh["cc_rig_optimize_ms"] > 0
