"""First-contact GPU check: blocks + solve parity of the HIP path against the oracle (diagnostic prints)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from camera_calibrator_amd import capi

def init_state(off, uv, xyz):
    K, q, t = po.zhang_init(off, uv, xyz)
    intr0 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    return intr0, q.astype(np.float64), t.astype(np.float64)

print("devices", capi.device_count())
for (F, M) in [(5, 100), (20, 88), (7, [3, 64, 65, 300, 1, 257, 128])]:
    off, uv, xyz = po.make_intrinsics_problem(F, M)
    intr0, q0, t0 = init_state(off, uv, xyz)
    prob = capi.IntrinsicsProblem(off, uv, xyz)
    prob.set_state(intr0, q0, t0)
    cost_g, blk_g = prob.eval()
    cost_o, blk_o = po.intrinsics_blocks(off, uv, xyz, intr0, q0, t0)
    scale = np.abs(blk_o).max(axis=(1, 2), keepdims=True)
    err = (np.abs(blk_g - blk_o) / scale).max()
    print(f"F={F} M={M}: cost gpu {cost_g:.12e} oracle {cost_o:.12e} block rel err {err:.3e}")
    if err > 1e-9:
        f = int(np.argmax((np.abs(blk_g - blk_o) / scale).max(axis=(1, 2))))
        np.set_printoptions(linewidth=250, precision=3)
        print("worst frame", f); print(blk_g[f][:4]); print(blk_o[f][:4])
    for graph in (0, 1):
        prob.reset()
        s = prob.solve(capi.default_options(use_graph=graph))
        ig, qg, tg = prob.get_state()
        io, qo, to, so = po.intrinsics_solve(off, uv, xyz, intr0, q0, t0)
        print(f"  graph={graph} gpu: it={s['iterations']} term={s['termination']} cost {s['final_cost']:.12e} t={s['seconds']*1e3:.2f} ms | oracle it={so['iterations']} term={so['termination']} cost {so['final_cost']:.12e}")
        print("   intr rel diff", np.abs(ig - io) / np.maximum(np.abs(io), 1e-12))
        print("   gpu costs", [f"{l['cost']:.10e}" for l in s['log']])
        print("   ora costs", [f"{l['cost']:.10e}" for l in so['log']])
    prob.close()

# C3
t0_ = time.time()
off, uv, xyz = po.make_intrinsics_problem(1000, 500)
intr0, q0, t0 = init_state(off, uv, xyz)
print("gen+init C3", time.time() - t0_)
prob = capi.IntrinsicsProblem(off, uv, xyz)
prob.set_state(intr0, q0, t0)
for rep in range(3):
    prob.reset()
    s = prob.solve()
    print(f"C3 solve: it={s['iterations']} term={s['termination']} cost {s['final_cost']:.10e} {s['seconds']*1e3:.3f} ms")
prob.reset()
s = prob.solve(capi.default_options(profile_kernels=1))
print("profile", s['kernel_ms'], s['kernel_launches'])
ig, _, _ = prob.get_state()
tt = time.time()
io, qo, to, so = po.intrinsics_solve(off, uv, xyz, intr0, q0, t0)
print("oracle C3", time.time() - tt, so['iterations'], so['termination'], so['final_cost'])
print("intr rel diff", np.abs(ig - io) / np.maximum(np.abs(io), 1e-12))
print(ig)
