"""One host thread driving several devices behind the drop-in surface (SURVEY.md 8(b) thread model):
cc_intrinsics_optimize_multi / cc_rig_optimize_multi and Calibrator::SetDevices / ExtrinsicsCalibrator::SetDevices.
On the one-GPU test box the devices are device 0 several times ("virtual devices"): same kernels, same mailbox
words, the shards' kernels concurrently resident and waiting for each other. The sharded solve must match the
one-device solve: identical accept sequence and iteration count, costs 1e-9 relative, parameters 1e-9, float32
class outputs within 1 ulp. (tests/multidevice_worker.py runs in its own process: see its docstring.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("part", ["intrinsics", "calibrator_class", "rig", "rig_class"])
def test_one_host_thread_drives_virtual_devices(part):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, os.path.join(HERE, "multidevice_worker.py"), part], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and " ok" in r.stdout, r.stdout[-3000:]
