"""bench.py's N > 1 path, rehearsed on ONE GPU (CC_BENCH_DEVICE=0 puts every rank on device 0): the launch line is the
driver's, the exchange goes through the mailboxes of two processes, and the line must carry the contract's fields for
N = 2 (weak scaling value + the strong-scaling leg). No physical multi-GPU run is claimed by this test."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra_env, frames, bare=False):
    env = dict(os.environ, CC_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               GPU_MAX_HW_QUEUES="8", **extra_env)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--frames", str(frames), "--points", "120"]
    if bare:   # the driver's one-GPU command shape with N = 2: no launcher in front, no WORLD_SIZE -- bench.py starts its ranks itself
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable] + bench
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + bench
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_rank_bench_line_over_the_mailbox_exchange():
    d = _run({"CC_EXCHANGE": "mailbox"}, 60)
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["config"]["exchange"] == "mailbox" and d["config"]["frames_total"] == 120
    assert d["value"] > 0 and abs(d["value"] - 2 * d["config"]["observations_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["converged"]["termination"] in ("FUNCTION", "PARAMETER", "GRADIENT")
    s = d["strong_scaling"]
    assert s["exchange"] == "mailbox" and 450 <= s["frames_per_gpu"] <= 550 and s["value"] > 0   # 1000 frames over 2 ranks
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None or d["n_gpus"] == 1   # rank-0, N = 1 only


@pytest.mark.gpu
def test_two_rank_bench_line_as_independent_replicas():
    d = _run({"CC_EXCHANGE": "replicas"}, 40)
    assert d["n_gpus"] == 2 and d["config"]["exchange"].startswith("none: independent replicas")
    assert d["config"]["parallelism"] == "independent replicas x2" and d["value"] > 0


@pytest.mark.gpu
def test_bare_command_starts_two_ranks_by_itself():
    d = _run({"CC_EXCHANGE": "mailbox"}, 40, bare=True)
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["config"]["exchange"] == "mailbox" and d["config"]["frames_total"] == 80
    assert [r["rank"] for r in d["config"]["ranks"]] == [0, 1] and d["value"] > 0 and d["strong_scaling"]["value"] > 0
