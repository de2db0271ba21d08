import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Load torch (and with it the ROCm runtime torch bundles) before libcc_hip.so, exactly as bench.py does:
# the C-ABI library then binds to the same HIP runtime / RCCL pair as torch.distributed.
try:
    import torch  # noqa: F401,E402
except Exception:  # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
