"""N>1 path on CPU: world_size-2 gloo processes run the frame-sharded LM (oracle kernels, the
product's cc_partition_frames for the split, torch.distributed for the exchange) and must
reproduce the single-process solve: same iteration count, same minimiser."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from camera_calibrator_amd import capi
    from oracle import pyoracle as po
    from tests.helpers import intrinsics_case
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = intrinsics_case(24, [40 + 7 * (i % 5) for i in range(24)])   # ragged frames
    first = capi.partition_frames(case["off"], world)
    f0, f1 = int(first[rank]), int(first[rank + 1])
    o0, o1 = int(case["off"][f0]), int(case["off"][f1])

    def allreduce(ctx, buf, n, op):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        a[:] = t.numpy()

    intr, q, t, s = po.intrinsics_solve(case["off"][f0:f1 + 1] - o0, case["uv"][o0:o1], case["xyz"][o0:o1],
                                        case["intr0"], case["q0"][f0:f1], case["t0"][f0:f1], allreduce=allreduce)
    qs = [None] * world
    dist.all_gather_object(qs, (f0, f1, q, t))
    if rank == 0:
        np.savez(out, intr=intr, iterations=s["iterations"], final_cost=s["final_cost"],
                 q=np.concatenate([x[2] for x in qs]), t=np.concatenate([x[3] for x in qs]),
                 first=first)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_matches_single_process(tmp_path, world):
    import torch.multiprocessing as tmp_mp
    from oracle import pyoracle as po
    from tests.helpers import intrinsics_case
    out = str(tmp_path / "sharded.npz")
    tmp_mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = np.load(out)
    case = intrinsics_case(24, [40 + 7 * (i % 5) for i in range(24)])
    intr, q, t, s = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    assert int(r["iterations"]) == s["iterations"]
    assert np.isclose(float(r["final_cost"]), s["final_cost"], rtol=1e-12)
    assert np.allclose(r["intr"], intr, rtol=1e-10, atol=1e-12)
    assert np.allclose(r["q"], q, atol=1e-10) and np.allclose(r["t"], t, atol=1e-10)


def _rig_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from camera_calibrator_amd import capi
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = po.rig_scenario(3, 30, 5)
    cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
    off = sc["frame_offsets"]
    first = capi.partition_frames(off, world)
    f0, f1 = int(first[rank]), int(first[rank + 1])
    o0, o1 = int(off[f0]), int(off[f1])

    def allreduce(ctx, buf, n, op):
        a = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        a[:] = t.numpy()

    r = po.rig_solve(3, off[f0:f1 + 1] - o0, sc["obs_cam"][o0:o1], sc["obs_world"][o0:o1], sc["obs_uv"][o0:o1],
                     sc["world_xyz"], cq, ct, sc["cam_frozen"], fq[f0:f1], ft[f0:f1], allreduce=allreduce,
                     cam_seen_global=[1, 1, 1])
    parts = [None] * world
    dist.all_gather_object(parts, (r[2], r[3], r[4]))
    if rank == 0:
        np.savez(out, cam_q=r[0], cam_t=r[1], frame_q=np.concatenate([p[0] for p in parts]),
                 frame_t=np.concatenate([p[1] for p in parts]), cost=np.concatenate([p[2] for p in parts]),
                 iterations=r[5]["iterations"], final_cost=r[5]["final_cost"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rig_solve_matches_single_process(tmp_path):
    import torch.multiprocessing as tmp_mp
    from oracle import pyoracle as po
    out = str(tmp_path / "rig_sharded.npz")
    tmp_mp.spawn(_rig_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    sc = po.rig_scenario(3, 30, 5)
    cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
    o = po.rig_solve(3, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct,
                     sc["cam_frozen"], fq, ft)
    assert int(r["iterations"]) == o[5]["iterations"]
    assert np.isclose(float(r["final_cost"]), o[5]["final_cost"], rtol=1e-11)
    assert np.allclose(r["cam_t"], o[1], atol=1e-10) and np.allclose(r["frame_t"], o[3], atol=1e-10)
    assert np.allclose(r["cost"], o[4], rtol=1e-8, atol=1e-16)
