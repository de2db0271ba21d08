#!/usr/bin/env python3
"""bench.py -- LM iterations/sec and residuals/sec of the HIP intrinsics bundle-adjustment path.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through
torch.distributed.run, one rank per GPU -- or bare: `python bench.py --gpus N` without WORLD_SIZE in the environment starts
its N ranks itself (launch_ranks) before anything touches the GPU.  A "step" is one Levenberg-Marquardt iteration (elimination
+ reduced solve + Jacobian sweep at the candidate + trust-region decision) of the single-camera
intrinsics problem of BASELINE.json configs[2] (1000 frames x 500 points) on synthetic
data_generator-style input already resident in HBM.  Steps are produced by complete solves from the
Zhang initialisation with the reference's own solver options (calibrator.cpp:314-321); a solve that
converges is followed by another one from the same initial state until exactly K steps have run.
The per-solve initial Jacobian sweep and host polls are inside the timed region and are NOT
counted as steps (conservative).  Before the W warmup steps the process runs 300 untimed solves so that the
GPU is at its steady clocks when the timed region starts (a cold process is ~4 % slower for its first few hundred
solves: with the driver's --steps 20 the whole timed region used to sit in that ramp); when K is not a
multiple of the solve's iteration count the one short solve comes first, the run ends on full solves.

For N>1 `value` is WEAK scaling: every rank owns 1000 frames x 500 points of one joint problem with N*1000
frames sharing one set of intrinsics; per LM iteration the ranks exchange 112 + 16 doubles (mailboxes in peer
HBM over xGMI, RCCL as the fallback).  The same line also carries `strong_scaling`: BASELINE.json configs[2]
exactly as written, the FIXED 1000 x 500 problem split N ways.

At N=1 the line additionally carries `configs` (BASELINE.json configs[3], configs[4]: the rig path at full
size, poses only = the reference's problem, with the shared-intrinsics extension and, at configs[4] size, with one set
of intrinsics per camera) and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_GPU = 1000
PTS_PER_FRAME = 500
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector = matrix peak (AMD datasheet)
FLOP_PER_OBS = 800.0       # ~0.8 kflop fp64 per observation and Jacobian sweep
# fp64 flops the rig sweeps EXECUTE per observation (counted in the ISA of the main loops): poses only, k_rig_sweep_adj --
# composed pose chain, projection, division, Huber test, two 7-entry rows, 42 FMAs of the 7-column Gram; with intrinsics,
# k_rig_sweep_adjk -- ~200 of pixel model and rows + 1024 of the 16 x 16 product of two rows on the matrix pipe
RIG_FLOP_PER_OBS = {"poses": 210.0, "shared_intrinsics": 1230.0, "per_camera_intrinsics": 1230.0}
# ... and k_rig_sweep_k2 (groups of >= 448 observations): ~200 of pixel model and rows + 2 x 210 of the lower-triangle Gram with the
# structural zeros of the pixel model skipped
RIG_FLOP_PER_OBS_K2 = 620.0


def algorithmic_bytes_sweep(n_obs, n_frames):
    # bytes_J = 20 N + 704 F
    return 20.0 * n_obs + 704.0 * n_frames


def algorithmic_bytes_rig_sweep(n_obs, n_world, n_frames, n_cams):
    # SURVEY.md 8(d), rig Jacobian sweep: 13 B per observation, 12 B per world point, per frame the pose (56 B) and
    # the block outputs (6x6 symmetric 21 + gradient 6 + 6x6 coupling per non-reference camera) doubles:
    # C4 8.1 MB, C5 120.6 MB
    return 13.0 * n_obs + 12.0 * n_world + n_frames * (56.0 + 8.0 * (27.0 + 36.0 * (n_cams - 1)))


def load_traffic(frames, points, kernel="k_intr_sweep"):
    """HBM bytes per launch of `kernel` from the committed PMC profile -- only when this run has the profiled shape."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if kernel.startswith("k_intr_persist"):
            t = t.get("intr_persist", {})
            key = "hbm_bytes_per_launch"
        else:
            key = "sweep_hbm_bytes_per_launch"
        shape = t.get("shape", {})
        if shape.get("frames") == frames and shape.get("points_per_frame") == points:
            return t.get(key)
    except Exception:
        pass
    return None


def load_pipe_busy(frames, points, kernel):
    """Matrix-pipe / vector-issue busy fractions of `kernel` from the committed counter pass (profiles/traffic.json, intr_persist.pipe_busy:
    rocprofv3 --pmc, scripts/pmc_busy_summary.py) -- only when this run has the profiled shape. Measured once per round on the same
    command; a reported figure next to the live one, not part of it."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f).get("intr_persist", {})
        shape = t.get("shape", {})
        if kernel.startswith("k_intr_persist") and shape.get("frames") == frames and shape.get("points_per_frame") == points:
            return t.get("pipe_busy")
    except Exception:
        pass
    return None


def load_rig_traffic(cams, frames, points, variant):
    """HBM bytes per launch of the rig sweep from the committed PMC profiles -- only for a profiled shape and variant."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        for e in [t.get("rig_sweep", {})] + list(t.get("rig_sweeps", [])):
            sh = e.get("shape", {})
            if (sh.get("cams"), sh.get("frames"), sh.get("points_per_frame"), sh.get("variant")) == (cams, frames, points, variant):
                return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def peer_access(torch, world):
    """hipDeviceCanAccessPeer for every pair of the node's first `world` devices (rank r uses device r): what the mailbox
    exchange needs. None on a single GPU."""
    if world < 2:
        return None
    n = min(world, torch.cuda.device_count())
    try:
        return [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
    except Exception as e:   # (reported, never fatal)
        return f"query failed: {e}"


def class_surface(frames, points):
    """What a drop-in user of the C++ class pays per call: Calibrator::Estimate at this size through the class (packing of the
    vector<Points2D> / vector<Points3D> arguments included), measured by the native program tests/cpp/test_dropin
    (--class-surface), a fresh Calibrator per call as the reference's workflow makes one. None when the program is not built."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--class-surface", str(frames), str(points), "20"], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stdout + r.stderr)[-400:]}
    except Exception as e:   # (reported, never fatal)
        return {"error": str(e)}


def class_surface_rig(cams, frames, points):
    """The same for the rig: ExtrinsicsCalibrator::Optimize at BASELINE configs[3] size through the class (flattening of the
    per-frame observation lists, cc_rig_optimize with its regrouping / upload / solve / per-observation costs, write-back)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--class-surface-rig", str(cams), str(frames), str(points), "8"], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stdout + r.stderr)[-400:]}
    except Exception as e:   # (reported, never fatal)
        return {"error": str(e)}


def usable_cores():
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, not the node's count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher. It starts
    N fresh children of this same file, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, the
    environment torch.distributed.run would give them), relays rank 0's JSON line and exits non-zero if any child does.
    The launcher never imports torch and never touches the GPU: the children are ordinary fresh processes, no exec from a
    process that has initialised HIP."""
    import socket
    import subprocess
    assert "torch" not in sys.modules, "the launcher must not load the GPU runtime"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what hipIpc / RCCL need on this driver
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None))
        # one rank that dies leaves its peers inside a rendezvous or a barrier: end them (exact PIDs) instead of waiting it out
        failed_at = None
        while any(q.poll() is None for q in procs):
            if failed_at is None and any(q.poll() not in (None, 0) for q in procs):
                failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > 10.0:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
            time.sleep(0.1)
        codes = [q.wait() for q in procs]
        out0.seek(0)
        out0 = out0.read()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if bad or len(lines) != 1:
        sys.stderr.write(f"bench.py launcher: ranks failed (rank, exit code): {bad}; rank 0 printed {len(lines)} JSON line(s)\n")
        return next((c for _, c in bad if c), 1) or 1
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (weak scaling)")
    ap.add_argument("--points", type=int, default=PTS_PER_FRAME)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the rig configurations (configs[3], configs[4])")
    ap.add_argument("--no-class-surface", action="store_true", help="skip the class-surface legs (they run as child processes of "
                    "their own: a rocprofv3 trace of this command then holds the bench loop's process only)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CC_BENCH_DEVICE" in os.environ:   # rehearsal on a one-GPU box: every rank on the same device
        local_rank = int(os.environ["CC_BENCH_DEVICE"])
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")

    import torch  # device/runtime plumbing: loads the ROCm runtime the library binds to

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from camera_calibrator_amd import capi

    def all_ok(flag):
        flags = [None] * world
        dist.all_gather_object(flags, bool(flag))
        return all(flags)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    opts = capi.default_options()
    if os.environ.get("CC_BENCH_GRAPH") == "0":   # experiment: eager launches instead of hipGraph replays
        opts.use_graph = 0

    class Leg:
        """One sharded intrinsics problem of F_total frames: this rank's handle, attached to its peers."""

        def __init__(self, F_total):
            # synthetic input: the product's DataGenerator harness (include/cc_harness.h, the reference's
            # data_generator.cpp without OpenCV) with the test_calibrator.cpp fixture constants; initial state from
            # the device Zhang initialisation (cc_zhang_init = Calibrator::Estimate before Optimize)
            self.F_total = F_total
            self.off, self.uv, self.xyz = capi.make_intrinsics_problem(F_total, args.points)
            K0, q0, t0 = capi.zhang_init(self.off, self.uv, self.xyz, device=local_rank)
            if dist is not None:   # every rank must start from the same bits: take rank 0's initial state
                init = [(K0, q0, t0) if rank == 0 else None]
                dist.broadcast_object_list(init, src=0)
                K0, q0, t0 = init[0]
            self.intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
            self.q0, self.t0 = q0.astype(np.float64), t0.astype(np.float64)
            first = capi.partition_frames(self.off, world)
            self.f0, self.f1 = int(first[rank]), int(first[rank + 1])
            self.o0, self.o1 = int(self.off[self.f0]), int(self.off[self.f1])
            self.my_off = self.off[self.f0:self.f1 + 1] - self.off[self.f0]
            self.n_obs_total = int(self.off[-1])
            self.prob = self._make()
            self.exchange = "none"
            self.exchange_errors = []          # why a route was not used (HIP / RCCL error strings), for the JSON line
            self.validation_us_per_iteration = None
            if world > 1:
                self._attach()

        def _make(self):
            p = capi.IntrinsicsProblem(self.my_off, self.uv[self.o0:self.o1], self.xyz[self.o0:self.o1], device=local_rank)
            p.set_state(self.intr0, self.q0[self.f0:self.f1], self.t0[self.f0:self.f1])
            return p

        def _attach(self):
            # The two per-iteration reductions are <= 1 KB: within a node they go through mailboxes in peer HBM
            # (cc_intrinsics_exchange_*, stores over xGMI, graph-captured); RCCL all-reduce is the fallback (and the
            # only choice beyond 8 ranks). CC_EXCHANGE=mailbox|rccl|replicas forces one.
            prob = self.prob
            want = os.environ.get("CC_EXCHANGE", "auto")
            if want in ("auto", "mailbox") and world <= 8:
                try:
                    mine = (True, prob.exchange_export())
                except capi.CcError as e:
                    mine = (False, str(e).encode())
                gathered = [None] * world
                dist.all_gather_object(gathered, mine)
                ok = all(g[0] for g in gathered)
                self.exchange_errors += [f"mailbox export, rank {r}: {g[1].decode(errors='replace')}" for r, g in enumerate(gathered) if not g[0]]
                if ok:
                    try:
                        prob.exchange_attach(rank, [g[1] for g in gathered])
                    except capi.CcError as e:
                        ok = False
                        self.exchange_errors.append(f"mailbox attach, rank {rank}: {e}")
                        print(f"[bench rank {rank}] mailbox attach failed: {e}", file=sys.stderr)
                ok = all_ok(ok)
                if ok:
                    try:  # one complete solve proves that every peer's posts arrive ...
                        prob.reset()
                        prob.solve(capi.default_options(), log_capacity=0)
                        prob.reset()
                        t_v = time.perf_counter()
                        chk = prob.solve(capi.default_options(), log_capacity=0)
                        self.validation_us_per_iteration = (time.perf_counter() - t_v) * 1e6 / max(1, chk["iterations"])
                        intr_chk, _, _ = prob.get_state()
                        local_cost, _ = prob.eval(want_blocks=False)
                    except capi.CcError as e:
                        ok = False
                        chk, intr_chk, local_cost = None, None, float("nan")
                        self.exchange_errors.append(f"mailbox validation solve, rank {rank}: {e}")
                        print(f"[bench rank {rank}] mailbox exchange failed: {e}", file=sys.stderr)
                    ok = all_ok(ok)
                    if ok:  # ... and that the exchanged sums are right: same bits everywhere, cost = sum of shards
                        parts = [None] * world
                        dist.all_gather_object(parts, (intr_chk.tobytes(), chk["final_cost"], local_cost))
                        same = all(p[0] == parts[0][0] and p[1] == parts[0][1] for p in parts)
                        total = sum(p[2] for p in parts)
                        ok = same and abs(total - chk["final_cost"]) <= 1e-9 * abs(total)
                        if not ok and rank == 0:
                            print(f"[bench] mailbox exchange gave inconsistent results (same={same}, "
                                  f"sum of shard costs {total!r} vs {chk['final_cost']!r})", file=sys.stderr)
                if ok:
                    self.exchange = "mailbox"
                elif want == "mailbox":
                    raise SystemExit("CC_EXCHANGE=mailbox but the mailbox exchange is not usable here")
                else:
                    dist.barrier()
                    prob.close()
                    self.prob = prob = self._make()
            if want == "replicas":   # rehearsal of the last resort below
                self.exchange = "none: independent replicas (forced by CC_EXCHANGE=replicas)"
                return
            if self.exchange == "none":
                ok, why = True, ""
                try:
                    uid = [capi.comm_get_unique_id() if rank == 0 else None]
                except capi.CcError as e:
                    uid, ok, why = [None], False, str(e)
                    self.exchange_errors.append(f"rccl unique id: {e}")
                dist.broadcast_object_list(uid, src=0)
                ok = all_ok(uid[0] is not None)
                if ok:
                    try:   # same acceptance test as for the mailboxes: one complete solve, identical bits on every rank
                        prob.comm_init(uid[0], rank, world)
                        prob.reset()
                        chk = prob.solve(capi.default_options(), log_capacity=0)
                        intr_chk, _, _ = prob.get_state()
                        mine = (intr_chk.tobytes(), chk["final_cost"])
                    except capi.CcError as e:
                        ok, why, mine = False, str(e), None
                        self.exchange_errors.append(f"rccl, rank {rank}: {e}")
                        print(f"[bench rank {rank}] RCCL exchange failed: {e}", file=sys.stderr)
                    parts = [None] * world
                    dist.all_gather_object(parts, mine)
                    ok = all(p is not None and p == parts[0] for p in parts)
                if ok:
                    self.exchange = "rccl"
                elif want == "rccl":
                    raise SystemExit(f"CC_EXCHANGE=rccl but the RCCL exchange is not usable here: {why}")
                else:
                    # last resort, so that the run still reports something honest: every rank solves its own shard as
                    # an independent problem (no joint intrinsics; DESIGN.md section 4 calls this "replicas")
                    if rank == 0:
                        print("[bench] neither exchange route works on this node: running independent replicas", file=sys.stderr)
                    dist.barrier()
                    prob.close()
                    self.prob = self._make()
                    self.exchange = "none: independent replicas (both exchange routes failed)"

        def run_steps(self, k):
            """Run exactly k LM iterations as consecutive complete solves; returns (solves, None). A solve takes
            `per` iterations (4 on this workload); when k is not a multiple, the short solve (its own max_iterations, i.e.
            an options upload in the library) comes FIRST, so that a run ends -- and the next one starts -- on the
            standard options: with the short solve last, the first timed solve paid for switching them back."""
            done, solves = 0, 0
            per = getattr(self, "_iters_per_solve", None)
            if per is None:   # one untimed full solve tells how many iterations a solve takes
                self.prob.reset()
                per = self._iters_per_solve = max(1, self.prob.solve_lean(opts))
            first = k % per
            while done < k:
                self.prob.reset()
                want = first if (done == 0 and first) else min(per, k - done)
                o = opts if want >= per else capi.default_options(max_iterations=want)
                its = self.prob.solve_lean(o)   # (the C call only: no Python summary dictionary inside the timed region)
                if its <= 0:
                    raise RuntimeError("solve made no progress")
                done += its
                solves += 1
            return solves, None

        def timed(self, steps, warmup):
            self.run_steps(warmup)
            barrier()
            t_start = time.perf_counter()
            solves, _ = self.run_steps(steps)
            barrier()
            elapsed = time.perf_counter() - t_start
            if dist is not None:
                tt = torch.tensor([elapsed], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elapsed = float(tt.item())
            return elapsed, solves

        def close(self):
            if dist is not None:
                dist.barrier()          # no rank frees a mailbox / communicator a peer may still use
            self.prob.close()
            if dist is not None:
                dist.barrier()

    leg = Leg(args.frames * world)
    prob = leg.prob

    # outside the timed region: one complete solve for reporting time-to-converge (the solve before it
    # pays for the one-off graph capture)
    prob.reset()
    prob.solve(opts, log_capacity=0)
    barrier()
    prob.reset()
    t_c0 = time.perf_counter()
    conv = prob.solve(opts, log_capacity=0)
    conv_ms = (time.perf_counter() - t_c0) * 1e3
    intr_final, _, _ = prob.get_state()
    # also outside the timed region (single GPU only): the one-shot C-ABI call a drop-in caller makes,
    # i.e. allocation + host->device upload of this rank's observations + solve + read-back
    e2e_ms = None
    if world == 1:
        one_shot = lambda: capi.intrinsics_optimize(leg.my_off, leg.uv, leg.xyz, leg.intr0, leg.q0, leg.t0, options=opts, log_capacity=0)
        one_shot()
        t_e = time.perf_counter()
        one_shot()
        e2e_ms = (time.perf_counter() - t_e) * 1e3

    # untimed: let the GPU reach its steady clocks before the contract's warmup + timed steps (a cold process runs
    # the first few hundred solves ~4 % slower; with --steps 20 the whole timed region would sit in that ramp)
    for _ in range(300):
        prob.reset()
        prob.solve_lean(opts)
    barrier()
    elapsed, solves = leg.timed(args.steps, args.warmup)

    # spread over individually timed solves (outside the contract's timed region; guards a short --steps run)
    per_iter_us = []
    for _ in range(20):
        prob.reset()
        barrier()
        t1 = time.perf_counter()
        s = prob.solve(opts, log_capacity=0)
        per_iter_us.append((time.perf_counter() - t1) * 1e6 / max(1, s["iterations"]))

    # ---- roofline of the dominant kernel, HIP events on the solver's stream ----
    # Persistent form (one launch = one complete solve: the only kernel of the path): algorithmic bytes of a launch =
    # evaluations it makes x (20 N + 704 F) (one fused residual + Jacobian sweep per evaluation),
    # launch time from cc_intrinsics_profile_solve. Two-kernel form: the Jacobian sweep kernel, as before.
    my_obs, my_frames = leg.o1 - leg.o0, leg.f1 - leg.f0
    bytes_sweep = algorithmic_bytes_sweep(my_obs, my_frames)
    form = prob.solver_form()
    prob.reset()
    prob.solve(opts, log_capacity=0)
    sweep_ms = prob.profile_sweep(100)          # k_intr_sweep alone (the two-kernel form's dominant kernel; reported either way)
    persist = None
    if form and world == 1:
        launch_ms, launch_sweeps = prob.profile_solve(opts, 50)
        persist = {"launch_ms": launch_ms, "evaluations_per_launch": launch_sweeps}
    prob.reset()
    prof = prob.solve(capi.default_options(profile_kernels=1), log_capacity=0)   # (profiled solves run the two-kernel form)
    if persist:
        dom_kernel, dom_ms, dom_bytes = f"k_intr_persist<{form}>", persist["launch_ms"], bytes_sweep * persist["evaluations_per_launch"]
        dom_flop = FLOP_PER_OBS * my_obs * persist["evaluations_per_launch"]
    else:
        dom_kernel, dom_ms, dom_bytes, dom_flop = "k_intr_sweep", sweep_ms, bytes_sweep, FLOP_PER_OBS * my_obs
    achieved_gbs = dom_bytes / (dom_ms * 1e-3) / 1e9
    fp64_tflops = dom_flop / (dom_ms * 1e-3) / 1e12

    # ---- strong scaling: BASELINE.json configs[2] as written, the fixed 1000 x 500 problem split over the ranks
    strong = None
    if world > 1:
        sleg = Leg(FRAMES_PER_GPU)
        s_elapsed, s_solves = sleg.timed(args.steps, args.warmup)
        s_sweep_ms = sleg.prob.profile_sweep(100)
        s_forms = [None] * world
        dist.all_gather_object(s_forms, {"rank": rank, "device": local_rank, "solver_form": sleg.prob.solver_form(), "exchange": sleg.exchange,
                                         "frames": sleg.f1 - sleg.f0, "solver_status": list(sleg.prob.solver_status())})
        strong = {
            "workload": f"fixed {FRAMES_PER_GPU} frames x {args.points} pts split over {world} GPUs",
            "ranks": s_forms,   # per rank: the form of the solver it ran (0 two kernels, 1 / 2 / 4 persistent) and the route of its exchange
            "value": 2.0 * sleg.n_obs_total * args.steps / s_elapsed, "unit": "residuals/s",
            "ms_per_step": s_elapsed / args.steps * 1e3, "lm_iterations_per_sec": args.steps / s_elapsed,
            "frames_per_gpu": sleg.f1 - sleg.f0, "exchange": sleg.exchange, "sweep_ms_rank0": s_sweep_ms,
        }
        sleg.close()

    rank_forms = None
    if world > 1:   # so that the first run on real hardware explains itself: every rank's solver form and exchange route
        rank_forms = [None] * world
        dist.all_gather_object(rank_forms, {"rank": rank, "device": local_rank, "solver_form": form, "exchange": leg.exchange,
                                            "frames": leg.f1 - leg.f0, "solver_status": list(prob.solver_status())})
    result = None
    if rank == 0:
        n_obs_total = leg.n_obs_total
        it_per_s = args.steps / elapsed
        res_per_s = 2.0 * n_obs_total * args.steps / elapsed
        result = {
            "metric": "LM residuals/sec (2*N_obs per LM iteration), single-camera intrinsics BA",
            "value": res_per_s,
            "unit": "residuals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE.json configs[2]: single-camera intrinsics, {args.frames} frames x "
                            f"{args.points} pts per GPU, radial-tangential distortion, Zhang init, "
                            f"reference solver options (calibrator.cpp:314-321)",
                "frames_total": leg.F_total, "points_per_frame": args.points,
                "observations_total": n_obs_total, "parallelism": (f"independent replicas x{world}" if world > 1 and leg.exchange.startswith("none") else f"frame-sharded x{world}"),
                "exchange": leg.exchange,
                "ranks": rank_forms,
                "exchange_errors": leg.exchange_errors,
                "exchange_validation_us_per_iteration": leg.validation_us_per_iteration,
                "peer_access": peer_access(torch, world),
            },
            "lm_iterations_per_sec": it_per_s,
            "solves_in_timed_region": solves,
            "iterations_per_solve": conv["iterations"],
            "time_to_converge_ms": conv_ms,
            "observations_per_sec": n_obs_total * args.steps / elapsed,
            "one_shot_ms_including_upload": e2e_ms,
            "class_surface": class_surface(args.frames, args.points) if world == 1 and not args.no_class_surface else None,
            "class_surface_rig": class_surface_rig(4, 400, 300) if world == 1 and not args.no_configs and not args.no_class_surface else None,
            "per_solve_us_per_iteration": {"n_solves": len(per_iter_us), "median": float(np.median(per_iter_us)),
                                           "min": float(np.min(per_iter_us)), "max": float(np.max(per_iter_us))},
            "converged": {"termination": conv["termination"], "final_cost": conv["final_cost"],
                          "intrinsics": [float(x) for x in intr_final]},
            "solver_form": ("persistent kernel, %d frame(s) per workgroup: one launch per solve" % form) if form else "two kernels per LM iteration",
            "roofline": {
                "kernel": dom_kernel,
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": load_traffic(my_frames, args.points, dom_kernel),
                "avg_launch_ms": dom_ms,
                "algorithmic_bytes_per_launch": dom_bytes,
                "evaluations_per_launch": persist["evaluations_per_launch"] if persist else 1,
                "note": "north_star asks for the HBM fraction; the kernel is bound by the fp64 pipe and by the latency of its "
                        "in-kernel hand-offs (roofline_fp64, DESIGN.md section 4)",
            },
            "roofline_fp64": {
                "kernel": dom_kernel, "bound": "fp64 mfma/valu", "achieved": fp64_tflops,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp64_tflops / FP64_PEAK_TFLOPS,
                "pmc": load_pipe_busy(my_frames, args.points, dom_kernel),
            },
            "sweep_kernel_alone": {
                "kernel": "k_intr_sweep", "avg_launch_ms": sweep_ms, "algorithmic_bytes_per_launch": bytes_sweep,
                "hbm_frac": bytes_sweep / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "fp64_frac": FLOP_PER_OBS * my_obs / (sweep_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "traffic": load_traffic(my_frames, args.points, "k_intr_sweep"),
            },
            "kernel_ms_per_launch": {
                k: (prof["kernel_ms"][k] / prof["kernel_launches"][k] if prof["kernel_launches"][k] else None)
                for k in prof["kernel_ms"]
            },
            "kernel_ms_labels": "the TWO-KERNEL form (what profile_kernels = 1 runs), eager launches with hipEvents around each: sweep = "
                                "k_intr_sweep, elim = k_intr_decide_elim (statistics + trust-region decision + pose elimination + 9x9 "
                                "solve step in one launch)",
        }
        if strong is not None:
            result["strong_scaling"] = strong
        if world == 1 and not args.no_configs:
            result["configs"] = rig_configs(capi, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle as po  # the ONLY use of oracle/ in this file: the cpu_baseline leg
            # CPU baseline: the oracle (a "port": exact-Schur fp64 restatement, analytic Jacobians), rebuilt on THIS
            # host with -O3 -march=native (the parity tests keep their -ffp-contract=off build),
            # same arrays, same options, 1 thread like the reference (Ceres num_threads default 1).
            flags, cpu_model = po.use_fast_build()
            oo = po.default_options()
            its, secs, n_solves = 0, 0.0, 0
            t_b = time.perf_counter()
            while time.perf_counter() - t_b < args.cpu_seconds:
                _, _, _, so = po.intrinsics_solve(leg.off, leg.uv, leg.xyz, leg.intr0, leg.q0, leg.t0, options=oo, log_capacity=0)
                its += so["iterations"]
                secs += so["seconds"]
                n_solves += 1
            cores = usable_cores()
            om = po.default_options(num_threads=cores)
            _, _, _, sm = po.intrinsics_solve(leg.off, leg.uv, leg.xyz, leg.intr0, leg.q0, leg.t0, options=om, log_capacity=0)
            cpu_res = 2.0 * n_obs_total * its / secs
            result["cpu_baseline"] = {
                "value": cpu_res, "unit": "residuals/s", "cores": 1, "kind": "port",
                "sample": f"{n_solves} complete solves ({its} LM iterations, {secs:.1f} s) of the same "
                          f"{leg.F_total}x{args.points} problem by oracle/liboracle_fast.so ({flags}) on {cpu_model}, 1 thread",
                "lm_iterations_per_sec": its / secs,
                "all_cores": {"cores": cores, "lm_iterations_per_sec": sm["iterations"] / sm["seconds"],
                              "value": 2.0 * n_obs_total * sm["iterations"] / sm["seconds"]},
            }
            result["speedup_vs_cpu_1thread"] = res_per_s / cpu_res
    leg.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def rig_configs(capi, device):
    """BASELINE.json configs[3] and configs[4] on one GPU: the rig path at full size (scenario of the reference's own
    rig test, include/cc_harness.h cc_rig_scenario), complete solves with the reference's options; per configuration
    the per-iteration time, the iteration count, the dominant kernel and its roofline fractions."""
    from camera_calibrator_amd import harness
    out = {}
    for name, (C_, F, M) in (("rig_c4", (4, 400, 300)), ("rig_c5", (8, 2000, 500))):
        for variant in ("poses", "shared_intrinsics") + (("per_camera_intrinsics",) if name == "rig_c5" else ()):
            if variant == "poses":
                sc = capi.rig_scenario(C_, F, M)
                cq, ct = capi.affine_to_qt(sc["cam_T"])
                fq, ft = capi.affine_to_qt(sc["frame_T"])
                n_obs, n_world = len(sc["obs_cam"]), len(sc["world_xyz"])
                prob = capi.RigProblem(C_, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"],
                                       sc["cam_frozen"], device=device)
            else:
                # extension: pixel observations of a rig through the fixture camera of test_calibrator.cpp:14-19
                # (camera_calibrator_amd/harness.py, the scenario of the parity tests)
                per_cam = variant == "per_camera_intrinsics"     # configs[4]: "full intrinsics+extrinsics co-optimisation"
                k = harness.rigk_case(C_, F, M, per_camera=per_cam)
                cq, ct, fq, ft = k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"]
                n_obs, n_world = len(k["obs_cam"]), len(k["world_xyz"])
                prob = capi.RigProblem(C_, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"],
                                       k["cam_frozen"], huber_a=0.0, device=device, with_intrinsics="per_camera" if per_cam else True)
                if per_cam:
                    for c in range(C_):
                        prob.set_camera_intrinsics(c, k["intr0"][c], 0)
                else:
                    prob.set_intrinsics(k["intr0"], 0)
            prob.set_state(cq, ct, fq, ft)
            form = prob.solver_form()
            o = capi.default_options(max_iterations=1000)
            s = prob.solve(o, log_capacity=0)
            ts = []
            for _ in range(3):
                prob.reset()
                t0 = time.perf_counter()
                s = prob.solve(o, log_capacity=0)
                ts.append(time.perf_counter() - t0)
            prob.reset()
            p = prob.solve(capi.default_options(max_iterations=1000, profile_kernels=1), log_capacity=0)
            prob.close()
            t_solve = float(np.median(ts))
            # per kernel ONE number: the mean over the launches that did work (the library keeps the launches of the last chunk
            # that return at once after the terminating decision apart: kernel_idle_*, cc_solver.h)
            per_launch = {k: (p["kernel_ms"][k] / p["kernel_launches"][k] if p["kernel_launches"][k] else None) for k in p["kernel_ms"]}
            sweep_ms = per_launch["sweep"]
            ab = algorithmic_bytes_rig_sweep(n_obs, n_world, F, C_)
            flop_per_obs = RIG_FLOP_PER_OBS_K2 if (variant != "poses" and n_obs / max(1, F * C_) >= 448) else RIG_FLOP_PER_OBS[variant]
            sweep_name = ("k_rig_sweep_frame (one workgroup per frame, a wave per (frame, camera) group)" if variant == "poses" else
                          ("k_rig_sweep_k2 (16-column Gram on plain FMAs, two waves per group, compact records)" if n_obs / max(1, F * C_) >= 448 else "k_rig_sweep_adjk (tiles)"))
            names = {"sweep": sweep_name, "decide": "k_rig_init", "elim": "k_rig_elim", "solve": "k_rig_reduce (column sums + reduced solve + pose update)",
                     "update": "k_rig_update", "reduce": "k_rig_reduce<2>", "allreduce": "ncclAllReduce"}
            # the kernel that takes the largest share of the profiled solve's kernel time -- NOT assumed to be the sweep: at
            # configs[3] the latency-bound reduce + solve + update launch is
            total_ms = sum(v for v in p["kernel_ms"].values() if v)
            dom = max((k for k in p["kernel_ms"] if p["kernel_launches"][k]), key=lambda k: p["kernel_ms"][k])
            dominant = {"kernel": names.get(dom, dom), "ms_per_full_launch": per_launch[dom], "share_of_kernel_time": p["kernel_ms"][dom] / total_ms}
            if dom == "sweep":
                dominant.update(bound="fp64 issue / hbm", hbm_frac=ab / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                fp64_frac=flop_per_obs * n_obs / (sweep_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)
            else:
                dominant.update(bound="latency", note="a chain of dependent steps on one workgroup (column sums, assembly, Cholesky of the "
                                "reduced system, pose update behind a flag): no bandwidth or flop roofline applies; stage times in "
                                "profiles/r04/rig_stage_marks.jsonl")
            if form:
                # the timed solves ran as ONE launch (lean persistent kernel + its control workgroup's launch); the per-kernel
                # figures below are the three-kernel form's, which is what a profiled solve runs
                dominant = {"kernel": "k_rig_persist_w (the whole solve in one launch; control workgroup: k_rig_persist_ctl)",
                            "ms_per_full_launch": t_solve * 1e3, "share_of_kernel_time": 1.0, "bound": "latency",
                            "note": "a round is a chain of dependent steps across workgroups (broadcast, pose update, sweep, elimination, "
                                    "column sums, reduced solve): no bandwidth or flop roofline applies; round timeline in "
                                    "profiles/r03/rig_persist_marks.jsonl"}
            out[f"{name}_{variant}"] = {
                "workload": f"rig {C_} cameras x {F} frames x {M} pts, {variant.replace('_', ' ')}"
                            + (" (= ExtrinsicsCalibrator::Optimize)" if variant == "poses" else " (extension, pixel observations)"),
                "observations": n_obs, "iterations": s["iterations"], "termination": s["termination"],
                "solve_ms": t_solve * 1e3, "ms_per_iteration": t_solve * 1e3 / max(1, s["iterations"]),
                "residuals_per_sec": 2.0 * n_obs * s["iterations"] / t_solve,
                "solver_form": {0: "three kernels per LM iteration", 2: "lean persistent kernel: one launch per solve"}[form],
                "dominant_kernel": dominant,
                "sweep_kernel": {"kernel": sweep_name, "ms_per_full_launch": sweep_ms,
                                 "hbm_frac": ab / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "fp64_frac": flop_per_obs * n_obs / (sweep_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                 "fp64_flop_executed_per_observation": flop_per_obs,
                                 "algorithmic_bytes_per_launch": ab, "traffic": load_rig_traffic(C_, F, M, variant)},
                "kernel_ms_per_full_launch_eager": per_launch,
                "idle_launches_of_the_profiled_solve": {k: v for k, v in p["kernel_idle_launches"].items() if v},
                "kernel_ms_labels": "(the THREE-KERNEL form: what profile_kernels = 1 runs) sweep = k_rig_sweep_frame (poses: per group the 7-column Gram of [J_cam r], per frame ONE "
                                    "assembly of the coupling blocks and the frame block through the groups' adjoints) / k_rig_sweep_k2 or k_rig_sweep_adjk (with intrinsics: sweep_kernel.kernel says which), decide = k_rig_init "
                                    "(once per solve), elim = k_rig_elim, solve = k_rig_reduce (column sums + reduced solve + pose update in one launch)",
                "final_cost": s["final_cost"],
            }
    return out


if __name__ == "__main__":
    main()
