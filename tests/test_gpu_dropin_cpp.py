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
