"""Generates the golden fixtures under tests/golden/.

The reference itself cannot be built or imported in the build container (Ceres/Eigen/OpenCV absent,
see DESIGN.md), and it ships no golden vectors, so these fixtures are inputs + outputs of the CPU
oracle (oracle/liboracle.so), committed so that (a) the oracle is regression-pinned and (b) the HIP
path is checked against fixed numbers on the GPU box.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import TIGHT, intrinsics_case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # C1: BASELINE.json configs[0] size, 20 frames x 88 points
    case = intrinsics_case(20, 88)
    cost0, blocks0 = po.intrinsics_blocks(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    intr, q, t, s = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"])
    intr_t, q_t, t_t, s_t = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"],
                                                options=po.default_options(**TIGHT))
    np.savez_compressed(
        os.path.join(OUT, "c1_intrinsics.npz"),
        off=case["off"], uv=case["uv"], xyz=case["xyz"], intr0=case["intr0"], q0=case["q0"], t0=case["t0"],
        cost0=cost0, blocks0_frame0=blocks0[0], blocks0_frame19=blocks0[19],
        intr_default=intr, cost_trace_default=np.array([l["cost"] for l in s["log"]]),
        accepted_default=np.array([l["accepted"] for l in s["log"]]), iterations_default=s["iterations"],
        intr_tight=intr_t, q_tight=q_t, t_tight=t_t, final_cost_tight=s_t["final_cost"])
    # rig: test_extrinsics_calibrator.cpp scenario downsized to 2 cams x 50 frames x 4 points
    sc = po.rig_scenario(2, 50, 4)
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    r = po.rig_solve(2, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct,
                     sc["cam_frozen"], fq, ft, options=po.default_options(max_iterations=1000, **{k: v for k, v in TIGHT.items() if k != "max_iterations"}))
    np.savez_compressed(
        os.path.join(OUT, "rig_2x50x4.npz"),
        frame_offsets=sc["frame_offsets"], obs_cam=sc["obs_cam"], obs_world=sc["obs_world"], obs_uv=sc["obs_uv"],
        world_xyz=sc["world_xyz"], cam_frozen=sc["cam_frozen"], cam_q0=cq, cam_t0=ct, frame_q0=fq, frame_t0=ft,
        cam_q=r[0], cam_t=r[1], frame_q=r[2], frame_t=r[3], obs_cost=r[4], final_cost=r[5]["final_cost"],
        initial_cost=r[5]["initial_cost"], cam_T_out=po.qt_to_affine(r[0], r[1]))
    # point kernels
    rng = np.random.default_rng(0)
    xy = rng.uniform(-0.7, 0.45, size=(256, 2)).astype(np.float32)
    uv = po.distort(po.FIXTURE_K, po.FIXTURE_DIST, xy)
    np.savez_compressed(os.path.join(OUT, "points.npz"), xy=xy, uv=uv, undist=po.undistort(po.FIXTURE_K, po.FIXTURE_DIST, uv))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
