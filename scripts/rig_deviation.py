"""How far is the HIP poses-only rig path (cc_rig_*, = ExtrinsicsCalibrator::Optimize) from the oracle, quantity by quantity,
with the approximate arithmetic of rounds 2-4 in the hot path (1/z from the hardware estimate + two Newton steps, Huber tail
from refined v_rsq_f64, Cholesky pivots / quaternion normalisation through rsqrt_pos)? One JSON line per case: max deviations
after identical default-option solves, and after tight solves with the step at which the oracle's cost changes drop into the
rounding of the cost sum (the bound that data gives for the tight iterate). The tolerances of tests/test_gpu_rig.py cite the
table this prints (profiles/r05/rig_deviation.jsonl).
    python scripts/rig_deviation.py [--c5]           # CC_LIB_PATH=scripts/ablate_build/libcc_exact.so for the exact build"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camera_calibrator_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import rig_outlier_case  # noqa: E402

TIGHT = dict(function_tolerance=1e-16, gradient_tolerance=1e-13, parameter_tolerance=1e-14, max_iterations=500)
THREADS = int(os.environ.get("ORACLE_THREADS", "1"))   # (1 = what the tests compare against; --c5 runs the oracle on 16)


def dev(g, o):
    n = min(len(g[5]["log"]), len(o[5]["log"]))
    rel_cost = max((abs(a["cost"] / b["cost"] - 1) for a, b in zip(g[5]["log"][:n], o[5]["log"][:n])), default=0.0)
    return dict(cam_q=float(np.abs(g[0] - o[0]).max()), cam_t=float(np.abs(g[1] - o[1]).max()),
                frame_q=float(np.abs(g[2] - o[2]).max()), frame_t=float(np.abs(g[3] - o[3]).max()),
                obs_cost_rel=float((np.abs(g[4] - o[4]) / np.maximum(np.abs(o[4]), 1e-300))[o[4] > 1e-14].max()),
                obs_cost_abs=float(np.abs(g[4] - o[4]).max()),
                log_cost_rel=float(rel_cost), final_cost_rel=float(abs(g[5]["final_cost"] / o[5]["final_cost"] - 1)),
                iterations=[g[5]["iterations"], o[5]["iterations"]],
                same_accepts=[l["accepted"] for l in g[5]["log"]] == [l["accepted"] for l in o[5]["log"]],
                terminations=[g[5]["termination"], o[5]["termination"]])


def noise_floor(log):
    """step norm of the first iteration whose cost change is below the rounding of the cost sum (tests/test_golden.py)"""
    for l in log:
        if abs(l["cost_change"]) < 1e-13 * l["cost"]:
            return float(l["step_norm"])
    return None


def case(name, cams, args, huber_a, tight=True):
    g = capi.rig_optimize(*args, huber_a=huber_a, options=capi.default_options(max_iterations=1000))
    o = po.rig_solve(*args, huber_a=huber_a, options=po.default_options(max_iterations=1000, num_threads=THREADS))
    tail = float((o[4] > 0.5 * huber_a * huber_a).mean()) if huber_a > 0 else 0.0
    out = dict(case=name, cams=cams, observations=int(len(args[2])), huber_tail_fraction=tail, lib=os.environ.get("CC_LIB_PATH", "default"),
               default=dev(g, o))
    if tight:
        g = capi.rig_optimize(*args, huber_a=huber_a, options=capi.default_options(**TIGHT))
        o = po.rig_solve(*args, huber_a=huber_a, options=po.default_options(num_threads=THREADS, **TIGHT))
        out["tight"] = dev(g, o)
        out["tight_noise_floor_step"] = noise_floor(o[5]["log"])
    print(json.dumps(out), flush=True)


def scenario(cams, frames, pts):
    sc = po.rig_scenario(cams, frames, pts)
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    return (cams, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)


if __name__ == "__main__":
    for sh in [(2, 1000, 4), (4, 40, 30), (8, 25, 70), (4, 400, 300), (23, 30, 6)]:
        case("rig_scenario %dx%dx%d" % sh, sh[0], scenario(*sh), capi.HUBER_A)
    for sh in [(3, 24, 8), (4, 60, 40)]:
        sc = rig_outlier_case(*sh)
        args = (sh[0], sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"],
                sc["cam_q0"], sc["cam_t0"], sc["cam_frozen"], sc["frame_q0"], sc["frame_t0"])
        case("rig_outlier_case %dx%dx%d" % sh, sh[0], args, capi.HUBER_A)
    if "--c5" in sys.argv:
        THREADS = 16
        case("rig_scenario 8x2000x500", 8, scenario(8, 2000, 500), capi.HUBER_A, tight=False)
