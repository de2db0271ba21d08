"""Where a round of the persistent rig kernels goes (default: the lean form; CC_RIG_PERSIST=1 CC_RIG_PERSIST_LEAN=0: the glued one) (timing-only build -DCC_RIG_PTIMING: wall-clock marks of worker 0 and of
the control workgroup in round 3, left in vec_stats). One JSON line."""
import ctypes as C, json, os, sys
sys.path.insert(0, ".")
import numpy as np
from camera_calibrator_amd import capi
from oracle import pyoracle as po
Cc, F, M = int(os.environ.get("C", 4)), int(os.environ.get("F", 400)), int(os.environ.get("M", 300))
sc = po.rig_scenario(Cc, F, M)
cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
prob = capi.RigProblem(Cc, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
prob.set_state(cq, ct, fq, ft)
s = prob.solve(capi.default_options(max_iterations=1000))
buf = np.zeros(64)
capi._check(capi.lib().cc_rig_debug_fetch(prob._h, b"vec_stats", capi._p(buf, C.c_double), C.c_int64(64)))
w = buf[8:19]; c = buf[40:46]
t0 = w[0]
names_w = ["round start (waits for B)", "B received", "poses updated", "groups swept", "statistics posted", "A received", "eliminated (after a decision broadcast: stale when the assumption held)", "row posted (same)", "column sums posted (end of the round's work)", "eliminated on the assumed decision", "assumed row posted"]
names_c = ["round start", "B posted", "statistics gathered", "A posted", "column sums gathered", "solve step done"]
print(json.dumps(dict(cams=Cc, frames=F, pts=M, form=prob.solver_form(), iterations=s["iterations"],
                      worker={n: round((x - t0) / 100.0, 2) for n, x in zip(names_w, w)},
                      control={n: round((x - t0) / 100.0, 2) for n, x in zip(names_c, c)})))
prob.close()
