"""Seeded random sweep of problem shapes through the HIP intrinsics path against the oracle: ragged
frames, frame counts around the kernels' internal tile sizes (16 frames per elimination block, 64
blocks, 1024 frames per statistics round trip), frozen-parameter masks and solver options."""
import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import intrinsics_case

pytestmark = pytest.mark.gpu


def _compare(case, mask=0, **kw):
    g = capi.intrinsics_optimize(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"],
                                 const_mask=mask, options=capi.default_options(**kw))
    okw = {k: v for k, v in kw.items() if k not in ("use_graph", "check_interval")}   # device-side knobs
    o = po.intrinsics_solve(case["off"], case["uv"], case["xyz"], case["intr0"], case["q0"], case["t0"],
                            const_mask=mask, options=po.default_options(**okw))
    assert g[3]["termination"] == o[3]["termination"] and g[3]["iterations"] == o[3]["iterations"]
    assert [l["accepted"] for l in g[3]["log"]] == [l["accepted"] for l in o[3]["log"]]
    assert np.allclose([l["cost"] for l in g[3]["log"]], [l["cost"] for l in o[3]["log"]], rtol=1e-9)
    assert np.allclose(g[0][:4], o[0][:4], rtol=1e-9) and np.allclose(g[0][4:], o[0][4:], atol=1e-8)
    assert np.allclose(g[1], o[1], atol=1e-8) and np.allclose(g[2], o[2], atol=1e-8)
    for j in range(9):
        if mask & (1 << j):
            assert g[0][j] == case["intr0"][j]


@pytest.mark.parametrize("frames", [15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 1100, 2500, 4100])
def test_frame_counts_around_the_tile_sizes(frames):
    _compare(intrinsics_case(frames, 12))


@pytest.mark.parametrize("seed", range(12))
def test_random_ragged_shapes_masks_and_options(seed):
    rng = np.random.default_rng(seed)
    frames = int(rng.integers(3, 90))
    pts = [int(x) for x in rng.choice([8, 9, 11, 16, 63, 64, 65, 100, 255, 256, 257, 300, 513], size=frames)]  # (4-7 point frames give near-degenerate Zhang poses: chaotic trajectories)
    case = intrinsics_case(frames, pts)
    mask = 0
    for j in rng.choice(np.arange(4, 9), size=int(rng.integers(0, 3)), replace=False):
        mask |= 1 << int(j)
    kw = {}
    if rng.uniform() < 0.3:
        kw["use_nonmonotonic_steps"] = 0
    if rng.uniform() < 0.3:
        kw["initial_radius"] = float(10.0 ** rng.uniform(1, 8))
    if rng.uniform() < 0.3:
        kw["check_interval"] = int(rng.integers(1, 7))
    if rng.uniform() < 0.3:
        kw["use_graph"] = 0
    _compare(case, mask, **kw)


# ---- rig path: frame counts around the 512 elimination blocks, camera counts up to the maximum, random visibility

def _rig_compare(cams, frames, pts, drop=0.0, seed=0):
    sc = po.rig_scenario(cams, frames, pts)
    if drop > 0.0:
        rng = np.random.default_rng(seed)
        keep = rng.uniform(size=len(sc["obs_cam"])) > drop
        off0 = sc["frame_offsets"]
        counts = [int(np.count_nonzero(keep[off0[f]:off0[f + 1]])) for f in range(frames)]
        sc = dict(sc, obs_cam=sc["obs_cam"][keep], obs_world=sc["obs_world"][keep], obs_uv=sc["obs_uv"][keep],
                  frame_offsets=np.concatenate([[0], np.cumsum(counts)]).astype(np.int64))
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    args = (cams, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
    kw = dict(max_iterations=25)           # the first 25 iterations: long tails at the noise floor are rounding-driven
    g = capi.rig_optimize(*args, options=capi.default_options(**kw))
    o = po.rig_solve(*args, options=po.default_options(**kw))
    assert g[5]["iterations"] == o[5]["iterations"] and g[5]["termination"] == o[5]["termination"]
    assert [l["accepted"] for l in g[5]["log"]] == [l["accepted"] for l in o[5]["log"]]
    assert np.allclose([l["cost"] for l in g[5]["log"]], [l["cost"] for l in o[5]["log"]], rtol=1e-9)
    for k in range(4):
        assert np.abs(g[k] - o[k]).max() < 1e-8
    assert np.allclose(g[4], o[4], rtol=1e-6, atol=1e-13)


@pytest.mark.parametrize("cams,frames,pts", [(2, 511, 4), (3, 512, 4), (2, 513, 5), (4, 1025, 4), (10, 40, 8), (9, 600, 3), (1, 30, 20)])
def test_rig_frame_and_camera_counts_around_the_limits(cams, frames, pts):
    _rig_compare(cams, frames, pts)


@pytest.mark.parametrize("seed", range(6))
def test_rig_random_visibility(seed):
    rng = np.random.default_rng(100 + seed)
    _rig_compare(int(rng.integers(2, 8)), int(rng.integers(20, 120)), int(rng.integers(6, 40)), drop=float(rng.uniform(0.1, 0.5)), seed=seed)
