"""The labelled experiment of scripts/dense_mfma (dense (6F+9)^2 Cholesky on the f64 matrix cores, NOT the product path)
must keep producing the product's step: it is the evidence behind DESIGN.md section 6."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "scripts", "dense_mfma")


@pytest.mark.gpu
@pytest.mark.parametrize("frames,points", [(20, 100), (150, 60)])
def test_dense_mfma_step_equals_the_schur_step(frames, points):
    subprocess.run(["bash", os.path.join(EXP, "build.sh")], check=True, timeout=600)
    out = subprocess.run([sys.executable, os.path.join(EXP, "run_dense.py"), str(frames), str(points)],
                         check=True, capture_output=True, text=True, timeout=600, env=dict(os.environ, REPS="1"))
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["n"] == 6 * frames + 9
    assert r["product_step_accepted"] is True
    assert r["step_rel_err_vs_host_schur"] < 1e-9       # dense step vs block-Schur step (numpy float64)
    assert r["state_rel_err_vs_product_iteration"] < 1e-9  # state after the dense step vs one iteration of the product
    assert r["dense_residual_rel"] < 1e-12
