"""The driver's bench.py contract, checked without a GPU: the committed sample line carries every field the
contract names, and bench.py refuses to run (loudly, no fallback) when there is no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_sample_line_has_every_contract_field():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_sample.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "residuals/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] in ("hbm", "mfma")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # (round 3: the dominant kernel is the persistent per-solve kernel; its observations are L2-resident after the first of
    # its five evaluations, so the counted HBM traffic may be well below the algorithmic bytes)
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.2
    assert r["kernel"].startswith("k_intr_persist") and r["evaluations_per_launch"] >= 1
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port") and c["cores"] == 1
    assert abs(d["value"] - 2 * d["config"]["observations_total"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    # round 2: the rig configurations ride in the same line, the CPU baseline says how it was built
    assert "-march=native" in c["sample"] and set(("rig_c4_poses", "rig_c5_poses")) <= set(d["configs"])
    # round 3: every rig configuration names the kernel that dominates ITS profiled solve, and keeps the sweep's fractions apart
    for v in d["configs"].values():
        assert set(("iterations", "ms_per_iteration", "dominant_kernel", "sweep_kernel")) <= set(v)
        assert set(("kernel", "ms_per_full_launch", "share_of_kernel_time", "bound")) <= set(v["dominant_kernel"])   # (round 5: one number per kernel, launches that did work)
        assert "ms_per_launch" not in v["dominant_kernel"] and "ms_per_full_launch" in v["sweep_kernel"]
        assert set(("hbm_frac", "fp64_frac", "algorithmic_bytes_per_launch")) <= set(v["sweep_kernel"])
    assert d["configs"]["rig_c4_poses"]["dominant_kernel"]["bound"] == "latency"


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return   # on the GPU box the real run is the driver's
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
    assert r.stdout.strip() == ""          # no JSON line from a run that measured nothing


def test_bare_multi_gpu_command_starts_its_ranks_before_any_gpu_call():
    """`python bench.py --gpus N` without WORLD_SIZE (what the driver's one-GPU command looks like with N = 8) must start N
    fresh rank processes itself -- with the environment torch.distributed.run would give them -- before torch or the HIP
    runtime is loaded in the launching process."""
    code = r"""
import json, os, sys
sys.argv = ["bench.py", "--gpus", "3", "--steps", "7", "--warmup", "2"]
os.environ.pop("WORLD_SIZE", None)
import subprocess
started = []
class FakePopen:
    def __init__(self, cmd, env=None, stdout=None, **kw):
        assert "torch" not in sys.modules and "camera_calibrator_amd.capi" not in sys.modules
        started.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        if env["RANK"] == "0":
            stdout.write(json.dumps({"metric": "x", "n_gpus": 3}) + chr(10)); stdout.flush()
        self.returncode = 0
    def poll(self): return 0
    def wait(self, timeout=None): return 0
    def kill(self): pass
subprocess.Popen = FakePopen
import bench
try:
    bench.main()
except SystemExit as e:
    rc = e.code
assert "torch" not in sys.modules, "launcher imported torch"
print(json.dumps({"rc": rc, "started": started}))
"""
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert json.loads(lines[0]) == {"metric": "x", "n_gpus": 3}          # rank 0's line, relayed
    rep = json.loads(lines[1])
    assert rep["rc"] == 0 and len(rep["started"]) == 3
    ports = set()
    for rank, (cmd, env) in enumerate(rep["started"]):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "3", "--steps", "7", "--warmup", "2"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank) and env["WORLD_SIZE"] == "3" and env["MASTER_ADDR"] == "127.0.0.1"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1


def test_bare_multi_gpu_command_fails_loudly_when_a_rank_fails():
    import torch
    if torch.cuda.is_available():
        return   # (on a GPU box the ranks would run: tests/test_gpu_bench_ranks.py covers that)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "needs a GPU" in r.stderr and "ranks failed" in r.stderr
    assert r.stdout.strip() == ""
