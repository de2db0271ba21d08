import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from camera_calibrator_amd import capi
np.set_printoptions(linewidth=220, precision=5)
Cn, F, M = 2, 6, 4
sc = po.rig_scenario(Cn, F, M)
cq, ct = po.affine_to_qt(sc["cam_T"]); fq, ft = po.affine_to_qt(sc["frame_T"])
prob = capi.RigProblem(Cn, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], sc["cam_frozen"])
prob.set_state(cq, ct, fq, ft)
s = prob.solve(capi.default_options(max_iterations=1, use_graph=0, check_interval=1))
def fetch(name, n):
    out = np.zeros(n)
    capi._check(capi.lib().cc_rig_debug_fetch(prob._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n)))
    return out
NG = F * Cn
gb = fetch("gblocks", 2 * NG * 256).reshape(2, NG, 16, 16)
ss = fetch("ss", 6 * Cn); sp = fetch("sp", F * 8).reshape(F, 8)[:, :6]
ds = fetch("ds", 6 * Cn)
# numpy model from oracle residual/Jacobian at x0
a = capi.HUBER_A
H = np.zeros((NG, 16, 16)); cost = 0
for f in range(F):
    for k in range(sc["frame_offsets"][f], sc["frame_offsets"][f + 1]):
        c = sc["obs_cam"][k]; X = sc["world_xyz"][sc["obs_world"][k]].astype(np.float64); uv = sc["obs_uv"][k].astype(np.float64)
        r, J = po.rig_residual(fq[f], ft[f], cq[c], ct[c], X, uv)
        sq = r @ r
        sr = 1.0 if sq <= a * a else np.sqrt(a / np.sqrt(sq))
        for i in range(2):
            v = np.zeros(16); v[:12] = J[i] * sr; v[12] = r[i] * sr
            if sc["cam_frozen"][c]: v[:6] = 0
            H[f * Cn + c] += np.outer(v, v)
print("block err (buffer 0 = x0)", np.abs(gb[0] - H).max() / np.abs(H).max())
hcc = np.zeros(6 * Cn)
for g in range(NG):
    c = g % Cn
    hcc[6 * c:6 * c + 6] += np.diag(H[g])[:6]
print("ss gpu", ss, "\nss ref", 1 / (1 + np.sqrt(hcc)))
hff = np.zeros((F, 6))
for g in range(NG):
    hff[g // Cn] += np.diag(H[g])[6:12]
print("sp err", np.abs(sp - 1 / (1 + np.sqrt(hff))).max())
# dense reference solve of the scaled damped system
n = 6 * Cn + 6 * F
Hd = np.zeros((n, n)); g_ = np.zeros(n)
for g in range(NG):
    c, f = g % Cn, g // Cn
    ic = np.arange(6 * c, 6 * c + 6); jf = 6 * Cn + np.arange(6 * f, 6 * f + 6)
    idx = np.concatenate([ic, jf])
    Hd[np.ix_(idx, idx)] += H[g][:12, :12]; g_[idx] += H[g][:12, 12]
scale = np.concatenate([1 / (1 + np.sqrt(hcc)), (1 / (1 + np.sqrt(hff))).ravel()])
Hs = Hd * scale[:, None] * scale[None, :]; gs = g_ * scale
D = np.clip(np.diag(Hs), 1e-6, 1e32) / 1e4
Aa = Hs + np.diag(D)
fixed = np.arange(6)  # cam 0 frozen
Aa[fixed, :] = 0; Aa[:, fixed] = 0; Aa[fixed, fixed] = 1; gs2 = gs.copy(); gs2[fixed] = 0
delta = -np.linalg.solve(Aa, gs2)
print("ds gpu", ds)
print("ds ref", delta[:6 * Cn])
frec = fetch("frec", F * 32).reshape(F, 32)
print("df gpu (unscaled) frame0", frec[0, 12:18], "\n ref", delta[6 * Cn:6 * Cn + 6] * scale[6 * Cn:6 * Cn + 6])
S = 6 * Cn; NP = S * (S + 1) // 2; PC = NP + 3 * S + 2
nblk = min(128, F)
part = fetch("partial", nblk * PC).reshape(nblk, PC)
tot = part.sum(axis=0)
print("partial b sum", tot[NP:NP + S])
print("partial hd sum", tot[NP + S:NP + 2 * S])
print("partial gs sum", tot[NP + 2 * S + 1:NP + 3 * S + 1], "fail", tot[NP + 2 * S], "gmax", part[:, -1].max())
print("ref gs", g_[:S])
# reference reduced system
Sfull = Hs[:S, :S] - sum(Hs[:S, S + 6 * f:S + 6 * f + 6] @ np.linalg.solve(Hs[S + 6 * f:S + 6 * f + 6, S + 6 * f:S + 6 * f + 6] + np.diag(D[S + 6 * f:S + 6 * f + 6]), Hs[S + 6 * f:S + 6 * f + 6, :S]) for f in range(F))
bfull = gs[:S] - sum(Hs[:S, S + 6 * f:S + 6 * f + 6] @ np.linalg.solve(Hs[S + 6 * f:S + 6 * f + 6, S + 6 * f:S + 6 * f + 6] + np.diag(D[S + 6 * f:S + 6 * f + 6]), gs[S + 6 * f:S + 6 * f + 6]) for f in range(F))
print("ref b", bfull)
iu = np.triu_indices(S)
print("S err", np.abs(tot[:NP] - Sfull[iu]).max())
print("debug shared_stats", fetch("shared_stats", 4))
print("S partial head", tot[:12])
