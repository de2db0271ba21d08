"""Stress of the mailbox exchange: W ranks on one GPU run ITER LM iterations with every tolerance switched off
(two exchanges per iteration, graph replay), several solves back to back; all ranks must stay bit-identical and
agree with a single-process solve. Usage: python scripts/exchange_stress.py [W=4] [ITER=1500]"""
import os, socket, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
rank, world, port, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
from camera_calibrator_amd import capi
off, uv, xyz = capi.make_intrinsics_problem(96, 40)
K0, q0, t0 = capi.zhang_init(off, uv, xyz)
intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
q0, t0 = q0.astype(np.float64), t0.astype(np.float64)
first = capi.partition_frames(off, world); f0, f1 = int(first[rank]), int(first[rank + 1]); o0, o1 = int(off[f0]), int(off[f1])
p = capi.IntrinsicsProblem(off[f0:f1 + 1] - off[f0], uv[o0:o1], xyz[o0:o1]); p.set_state(intr0, q0[f0:f1], t0[f0:f1])
h = [None] * world; dist.all_gather_object(h, p.exchange_export()); p.exchange_attach(rank, h)
opt = capi.default_options(max_iterations=iters, function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0, min_radius=0.0)
out = []
for rep in range(3):
    p.reset(); s = p.solve(opt, log_capacity=0); out.append((s["iterations"], s["final_cost"], p.get_state()[0].tobytes()))
allr = [None] * world; dist.all_gather_object(allr, out)
ok = all(a == allr[0] for a in allr)
if rank == 0: print("ranks identical:", ok, "iterations", [o[0] for o in out], "cost", out[0][1], flush=True)
dist.barrier(); p.close(); dist.barrier()
sys.exit(0 if ok else 3)
''' % ROOT

def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), str(port), str(iters)]) for r in range(world)]
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill(); rc |= 1
    print("stress", "PASSED" if rc == 0 else "FAILED", f"({world} ranks, {iters} iterations x 3 solves)")
    sys.exit(rc)

if __name__ == "__main__":
    main()
