#!/usr/bin/env python3
"""bench.py -- LM iterations/sec and residuals/sec of the HIP intrinsics bundle-adjustment path.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through
torch.distributed.run, one rank per GPU.  A "step" is one Levenberg-Marquardt iteration (elimination
+ reduced solve + Jacobian sweep at the candidate + trust-region decision) of the single-camera
intrinsics problem of BASELINE.json configs[2] (1000 frames x 500 points) on synthetic
data_generator-style input already resident in HBM.  Steps are produced by complete solves from the
Zhang initialisation with the reference's own solver options (calibrator.cpp:314-321); a solve that
converges is followed by another one from the same initial state until exactly K steps have run.
The per-solve initial Jacobian sweep and host polls are inside the timed region and are NOT
counted as steps (conservative).

For N>1 every rank owns 1000 frames x 500 points (weak scaling): the global problem has N*1000
frames sharing one set of intrinsics; ranks all-reduce (RCCL) 2 x 64 doubles per LM iteration.
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


def algorithmic_bytes_sweep(n_obs, n_frames):
    # bytes_J = 20 N + 704 F
    return 20.0 * n_obs + 704.0 * n_frames


def load_traffic():
    """HBM bytes per sweep launch from the committed PMC profile (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("sweep_hbm_bytes_per_launch")
    except Exception:
        return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU")
    ap.add_argument("--points", type=int, default=PTS_PER_FRAME)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CC_BENCH_DEVICE" in os.environ:   # rehearsal on a one-GPU box: every rank on the same device
        local_rank = int(os.environ["CC_BENCH_DEVICE"])
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")

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

    # ---- synthetic input: the product's DataGenerator harness (include/cc_harness.h, the reference's
    # data_generator.cpp without OpenCV) with the test_calibrator.cpp fixture constants; initial state
    # from the device Zhang initialisation (cc_zhang_init = Calibrator::Estimate before Optimize)
    F_total = args.frames * world
    off, uv, xyz = capi.make_intrinsics_problem(F_total, args.points)
    K0, q0, t0 = capi.zhang_init(off, uv, xyz, device=local_rank)
    if dist is not None:   # every rank must start from the same bits: take rank 0's initial state
        init = [(K0, q0, t0) if rank == 0 else None]
        dist.broadcast_object_list(init, src=0)
        K0, q0, t0 = init[0]
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    q0 = q0.astype(np.float64)
    t0 = t0.astype(np.float64)
    first = capi.partition_frames(off, world)
    f0, f1 = int(first[rank]), int(first[rank + 1])
    o0, o1 = int(off[f0]), int(off[f1])
    my_off = off[f0:f1 + 1] - off[f0]
    def make_problem():
        p = capi.IntrinsicsProblem(my_off, uv[o0:o1], xyz[o0:o1], device=local_rank)
        p.set_state(intr0, q0[f0:f1], t0[f0:f1])
        return p

    def all_ok(flag):
        flags = [None] * world
        dist.all_gather_object(flags, bool(flag))
        return all(flags)

    prob = make_problem()
    exchange = "none"
    if world > 1:
        # The two per-iteration reductions are <= 1 KB: within a node they go through mailboxes in peer
        # HBM (cc_intrinsics_exchange_*, stores over xGMI, graph-captured); RCCL all-reduce is the
        # fallback (and the only choice beyond 8 ranks). CC_EXCHANGE=mailbox|rccl forces one.
        want = os.environ.get("CC_EXCHANGE", "auto")
        if want in ("auto", "mailbox") and world <= 8:
            try:
                mine = (True, prob.exchange_export())
            except capi.CcError as e:
                mine = (False, str(e).encode())
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            ok = all(g[0] for g in gathered)
            if ok:
                try:
                    prob.exchange_attach(rank, [g[1] for g in gathered])
                except capi.CcError as e:
                    ok = False
                    print(f"[bench rank {rank}] mailbox attach failed: {e}", file=sys.stderr)
            ok = all_ok(ok)
            if ok:
                try:  # one complete solve proves that every peer's posts arrive ...
                    prob.reset()
                    chk = prob.solve(capi.default_options(), log_capacity=0)
                    intr_chk, _, _ = prob.get_state()
                    local_cost, _ = prob.eval(want_blocks=False)
                except capi.CcError as e:
                    ok = False
                    chk, intr_chk, local_cost = None, None, float("nan")
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
                exchange = "mailbox"
            elif want == "mailbox":
                raise SystemExit("CC_EXCHANGE=mailbox but the mailbox exchange is not usable here")
            else:
                dist.barrier()
                prob.close()
                prob = make_problem()
        if exchange == "none":
            uid = [capi.comm_get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            prob.comm_init(uid[0], rank, world)
            exchange = "rccl"

    n_obs_total = int(off[-1])
    opts = capi.default_options()

    def run_steps(k):
        """Run exactly k LM iterations as consecutive complete solves; returns per-solve stats."""
        done, solves, last = 0, 0, None
        while done < k:
            prob.reset()
            remaining = k - done
            o = opts if remaining >= opts.max_iterations else capi.default_options(max_iterations=remaining)
            s = prob.solve(o, log_capacity=0)
            if s["iterations"] <= 0:
                raise RuntimeError(f"solve made no progress: {s}")
            done += s["iterations"]
            solves += 1
            last = s
        return solves, last

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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
        capi.intrinsics_optimize(my_off, uv[o0:o1], xyz[o0:o1], intr0, q0[f0:f1], t0[f0:f1], options=opts, log_capacity=0)
        t_e = time.perf_counter()
        capi.intrinsics_optimize(my_off, uv[o0:o1], xyz[o0:o1], intr0, q0[f0:f1], t0[f0:f1], options=opts, log_capacity=0)
        e2e_ms = (time.perf_counter() - t_e) * 1e3

    run_steps(args.warmup)
    barrier()
    t_start = time.perf_counter()
    solves, last = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- roofline of the dominant kernel (Jacobian sweep), HIP events on the solver's stream ----
    prob.reset()
    prob.solve(opts, log_capacity=0)
    sweep_ms = prob.profile_sweep(100)
    prob.reset()
    prof = prob.solve(capi.default_options(profile_kernels=1), log_capacity=0)
    my_obs, my_frames = o1 - o0, f1 - f0
    bytes_sweep = algorithmic_bytes_sweep(my_obs, my_frames)
    achieved_gbs = bytes_sweep / (sweep_ms * 1e-3) / 1e9
    fp64_tflops = FLOP_PER_OBS * my_obs / (sweep_ms * 1e-3) / 1e12

    result = None
    if rank == 0:
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
                "frames_total": F_total, "points_per_frame": args.points,
                "observations_total": n_obs_total, "parallelism": f"frame-sharded x{world}", "exchange": exchange,
            },
            "lm_iterations_per_sec": it_per_s,
            "solves_in_timed_region": solves,
            "iterations_per_solve": conv["iterations"],
            "time_to_converge_ms": conv_ms,
            "observations_per_sec": n_obs_total * args.steps / elapsed,
            "one_shot_ms_including_upload": e2e_ms,
            "converged": {"termination": conv["termination"], "final_cost": conv["final_cost"],
                          "intrinsics": [float(x) for x in intr_final]},
            "roofline": {
                "kernel": "k_intr_sweep",
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": load_traffic(),
                "avg_launch_ms": sweep_ms,
                "algorithmic_bytes_per_launch": bytes_sweep,
            },
            "roofline_fp64": {
                "kernel": "k_intr_sweep", "bound": "fp64 mfma/valu", "achieved": fp64_tflops,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp64_tflops / FP64_PEAK_TFLOPS,
            },
            "kernel_ms_per_launch": {
                k: (prof["kernel_ms"][k] / prof["kernel_launches"][k] if prof["kernel_launches"][k] else None)
                for k in prof["kernel_ms"]
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle as po  # the ONLY use of oracle/ in this file: the cpu_baseline leg
            # CPU baseline: the oracle (a "port": exact-Schur fp64 restatement, analytic Jacobians),
            # same arrays, same options, 1 thread like the reference (Ceres num_threads default 1).
            oo = po.default_options()
            its, secs, n_solves = 0, 0.0, 0
            t_b = time.perf_counter()
            while time.perf_counter() - t_b < args.cpu_seconds:
                _, _, _, so = po.intrinsics_solve(off, uv, xyz, intr0, q0, t0, options=oo, log_capacity=0)
                its += so["iterations"]
                secs += so["seconds"]
                n_solves += 1
            cores = usable_cores()
            om = po.default_options(num_threads=cores)
            _, _, _, sm = po.intrinsics_solve(off, uv, xyz, intr0, q0, t0, options=om, log_capacity=0)
            cpu_res = 2.0 * n_obs_total * its / secs
            result["cpu_baseline"] = {
                "value": cpu_res, "unit": "residuals/s", "cores": 1, "kind": "port",
                "sample": f"{n_solves} complete solves ({its} LM iterations, {secs:.1f} s) of the same "
                          f"{F_total}x{args.points} problem by oracle/liboracle.so, 1 thread",
                "lm_iterations_per_sec": its / secs,
                "all_cores": {"cores": cores, "lm_iterations_per_sec": sm["iterations"] / sm["seconds"],
                              "value": 2.0 * n_obs_total * sm["iterations"] / sm["seconds"]},
            }
            result["speedup_vs_cpu_1thread"] = res_per_s / cpu_res
    if dist is not None:
        dist.barrier()          # no rank frees a mailbox / communicator a peer may still use
    prob.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
