"""The two forms of the intrinsics solver against the same parity suite.

cc_intrinsics_solve runs the persistent per-solve kernel whenever every frame fits a resident workgroup -- with the
FEWEST frames per workgroup that do, i.e. one frame per workgroup for every problem the parity tests are small enough to
check against the oracle. So the suite runs again with two and four frames per workgroup forced (teams idle where the
frame count is no multiple, frames of different lengths sharing a workgroup's barriers, empty frames next to full ones),
and once with the persistent kernel switched off: the two-kernels-per-iteration form that large problems, profiled solves
and the RCCL route use. Same bar as tests/test_gpu_intrinsics.py in every form."""
import os
import subprocess
import sys

import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import intrinsics_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_gpu_intrinsics.py", "tests/test_gpu_random.py", "tests/test_gpu_lm_branches.py", "tests/test_gpu_edge_inputs.py",
         "tests/test_golden.py"]


@pytest.mark.parametrize("env", [{"CC_INTR_PERSIST_TEAMS": "2"}, {"CC_INTR_PERSIST_TEAMS": "4"}, {"CC_INTR_PERSIST": "0"}],
                         ids=["two_frames_per_workgroup", "four_frames_per_workgroup", "two_kernel_form"])
def test_the_intrinsics_parity_suite_in_every_form_of_the_solver(env):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", *FILES], cwd=ROOT,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]


def test_solver_form_follows_the_frame_count():
    """One frame per workgroup while the frames fit the compute units, then two, then four; beyond four per unit, tiled
    frames or CC_INTR_PERSIST=0: the two-kernel form (0)."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for frames, want in ((3, 1), (cus - 1, 1), (cus, 2), (2 * (cus - 1), 2), (2 * cus, 4), (4 * (cus - 1), 4), (4 * cus, 0)):
        off = (np.arange(frames + 1) * 4).astype(np.int64)
        p = capi.IntrinsicsProblem(off, np.zeros((4 * frames, 2), np.float32), np.zeros((4 * frames, 3), np.float32))
        assert p.solver_form() == want, (frames, p.solver_form(), want)
        p.close()


def test_both_forms_agree_with_each_other_and_with_the_oracle_at_configs2_size():
    """BASELINE.json configs[2] (1000 x 500: four frames per workgroup, 250 workers + control): same accept / reject
    sequence and costs as the oracle, and as the two-kernel form in a process of its own (CC_INTR_PERSIST is read when a
    handle is created)."""
    c = intrinsics_case(1000, 500)
    prob = capi.IntrinsicsProblem(c["off"], c["uv"], c["xyz"])
    prob.set_state(c["intr0"], c["q0"], c["t0"])
    assert prob.solver_form() == 4
    s = prob.solve()
    ig, qg, tg = prob.get_state()
    s2 = prob.solve()                                   # continues from the accepted point: nothing left to do
    prob.close()
    io, qo, to, so = po.intrinsics_solve(c["off"], c["uv"], c["xyz"], c["intr0"], c["q0"], c["t0"], options=po.default_options(num_threads=8))
    assert s["iterations"] == so["iterations"] and s["termination"] == so["termination"]
    assert [l["accepted"] for l in s["log"]] == [l["accepted"] for l in so["log"]]
    assert np.allclose([l["cost"] for l in s["log"]], [l["cost"] for l in so["log"]], rtol=1e-9)
    assert np.allclose([l["gradient_max_norm"] for l in s["log"]], [l["gradient_max_norm"] for l in so["log"]], rtol=1e-6)
    assert np.allclose(ig[:4], io[:4], rtol=1e-9) and np.allclose(ig[4:], io[4:], atol=1e-9)
    assert np.abs(qg - qo).max() < 1e-9 and np.abs(tg - to).max() < 1e-9
    assert s2["iterations"] <= 1
