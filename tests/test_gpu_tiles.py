"""Frame-tile mode of the intrinsics sweep (several workgroups per frame when the frames alone cannot fill the chip:
a shard of BASELINE.json configs[2] on 4-8 GPUs has 125-250 frames). Same parity bar as tests/test_gpu_intrinsics.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import block_rel_err, intrinsics_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_few_long_frames_are_tiled_and_match_the_oracle():
    """24 frames x 1500 points: three tiles per frame by default (ceil(1500 / 512), 24 * 3 workgroups <= 256 CUs)."""
    c = intrinsics_case(24, 1500)
    prob = capi.IntrinsicsProblem(c["off"], c["uv"], c["xyz"])
    prob.set_state(c["intr0"], c["q0"], c["t0"])
    cost_g, blocks_g = prob.eval()
    cost_o, blocks_o = po.intrinsics_blocks(c["off"], c["uv"], c["xyz"], c["intr0"], c["q0"], c["t0"])
    assert block_rel_err(blocks_g, blocks_o) < 1e-12 and np.isclose(cost_g, cost_o, rtol=1e-13)
    s = prob.solve()
    ig, qg, tg = prob.get_state()
    prob.close()
    io, qo, to, so = po.intrinsics_solve(c["off"], c["uv"], c["xyz"], c["intr0"], c["q0"], c["t0"])
    assert s["iterations"] == so["iterations"] and s["termination"] == so["termination"]
    assert [l["accepted"] for l in s["log"]] == [l["accepted"] for l in so["log"]]
    assert np.allclose([l["cost"] for l in s["log"]], [l["cost"] for l in so["log"]], rtol=1e-9)
    assert np.allclose(ig[:4], io[:4], rtol=1e-9) and np.allclose(ig[4:], io[4:], atol=1e-9)
    assert np.abs(qg - qo).max() < 1e-9 and np.abs(tg - to).max() < 1e-9


@pytest.mark.parametrize("tiles", ["2", "5"])
def test_the_intrinsics_parity_suite_with_forced_tiles(tiles):
    """Every intrinsics parity test (model blocks, trajectories, LM branches, ragged and degenerate inputs) again with
    the frames cut into tiles, including tiles that are empty for short frames."""
    env = dict(os.environ, CC_SWEEP_TILES=tiles)
    files = ["tests/test_gpu_intrinsics.py", "tests/test_gpu_random.py", "tests/test_gpu_lm_branches.py", "tests/test_gpu_edge_inputs.py",
             "tests/test_golden.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", *files], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
