"""The forms of the rig sweep give the same numbers.

Per (frame, camera) group only the columns [J_cam(6) r (J_k(9))] are accumulated and the frame blocks follow from the group's
adjoint, J_frame = J_cam M (cc_rig.hip). The group form writes a 16 x 16 tile per group (k_rig_sweep_adj / k_rig_sweep_adjk),
the frame form (k_rig_sweep_frame) and the FMA sweep with intrinsics (k_rig_sweep_k2) write compact records. Compared here: what
the elimination reads after the initial evaluation of a solve, entry by entry, and complete solves against the oracle under
either. (The first formulation -- all 13 / 22 columns of a row through the matrix pipe, CC_RIG_SWEEP_MFMA=1 -- was deleted in
round 5 together with the tests that compared it with the adjoint form.)"""
import ctypes as C

import numpy as np
import pytest

from camera_calibrator_amd import capi
from oracle import pyoracle as po
from tests.helpers import rigk_case
from tests.test_gpu_rig import _assert_same, _both
from tests.test_gpu_rigk import _assert_same as _assert_same_k, _both as _both_k

pytestmark = pytest.mark.gpu


def _fetch(prob, name, n):
    buf = np.zeros(n)
    capi._check(capi.lib().cc_rig_debug_fetch(prob._h, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n)))
    return buf


def _group_count(sc_or_k, cams):
    off, cam = np.asarray(sc_or_k["frame_offsets"]), np.asarray(sc_or_k["obs_cam"])
    return sum(len(np.unique(cam[off[f]:off[f + 1]])) for f in range(len(off) - 1))


def _blocks_poses_only(monkeypatch, sc, cams, frozen, huber_a):
    monkeypatch.setenv("CC_RIG_SWEEP_FRAME", "0")   # (the 16 x 16 tiles: the frame form of the sweep never writes them, see below)
    monkeypatch.setenv("CC_RIG_PERSIST", "0")    # (the group blocks are read back from global memory: the lean persistent kernel keeps them in LDS)
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    prob = capi.RigProblem(cams, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], frozen, huber_a=huber_a)
    prob.set_state(cq, ct, fq, ft)
    cost = prob.solve(capi.default_options(max_iterations=1))["initial_cost"]    # (cc_rig_eval does not run the sweep)
    blocks = _fetch(prob, "gblocks", _group_count(sc, cams) * 256).reshape(-1, 16, 16)   # buffer 0: the initial point
    prob.close()
    return cost, blocks


@pytest.mark.parametrize("cams,frames,pts,huber_a", [(3, 30, 150, capi.HUBER_A), (4, 25, 70, 0.0), (2, 40, 5, capi.HUBER_A)])
def test_group_tiles_have_the_expected_structure(monkeypatch, cams, frames, pts, huber_a):
    sc = po.rig_scenario(cams, frames, pts)
    frozen = np.array(sc["cam_frozen"], dtype=np.uint8).copy()
    c0, b0 = _blocks_poses_only(monkeypatch, sc, cams, frozen, huber_a)
    assert c0 >= 0 and np.abs(b0).max() > 0   # (huber_a = 0: every block sits in the tail with weight and cost zero at a = 0)
    assert np.array_equal(b0[:, 13:, :], np.zeros_like(b0[:, 13:, :])) and np.array_equal(b0[:, :, 13:], np.zeros_like(b0[:, :, 13:]))
    assert np.abs(b0 - np.transpose(b0, (0, 2, 1))).max() <= 1e-12 * np.abs(b0).max()
    # frozen camera (camera 0 of the scenario): its own rows and columns are zero, its frames' blocks are not
    assert frozen[0] == 1
    zero_cam = np.abs(b0[:, :6, :]).max(axis=(1, 2)) == 0
    assert zero_cam.any() and np.all(np.diagonal(b0[zero_cam][:, 6:12, 6:12], axis1=1, axis2=2) > 0)


@pytest.mark.parametrize("nw", [1, 2, 4, 8])
@pytest.mark.parametrize("cams,frames,pts,huber_a", [(3, 30, 150, capi.HUBER_A), (4, 25, 70, 0.0), (2, 40, 5, capi.HUBER_A), (11, 12, 20, capi.HUBER_A)])
def test_frame_form_records_equal_the_tiles(monkeypatch, nw, cams, frames, pts, huber_a):
    """The FRAME form of the sweep (k_rig_sweep_frame, the default of the three-kernel path since round 4) writes per group
    [G7 (28) | T = G_cc M (36)] and per frame [H_ff (21) | g_f (6)] instead of the 16 x 16 tile N^T G7 N: every one of these
    numbers is an entry (or, for the frame record, a sum over the frame's groups of entries) of the tile the group form
    writes -- compared here entry by entry, with every wave count of the frame workgroup (more groups per frame than one
    assembly pass of eight in the last shape), together with the frame's cost row."""
    sc = po.rig_scenario(cams, frames, pts)
    frozen = np.array(sc["cam_frozen"], dtype=np.uint8).copy()
    c0, b0 = _blocks_poses_only(monkeypatch, sc, cams, frozen, huber_a)
    monkeypatch.setenv("CC_RIG_SWEEP_FRAME", "1")
    monkeypatch.setenv("CC_RIG_FRAME_WAVES", str(nw))
    cq, ct = po.affine_to_qt(sc["cam_T"])
    fq, ft = po.affine_to_qt(sc["frame_T"])
    prob = capi.RigProblem(cams, sc["frame_offsets"], sc["obs_cam"], sc["obs_world"], sc["obs_uv"], sc["world_xyz"], frozen, huber_a=huber_a)
    prob.set_state(cq, ct, fq, ft)
    c1 = prob.solve(capi.default_options(max_iterations=1))["initial_cost"]
    ng = _group_count(sc, cams)
    rec = _fetch(prob, "gcomp", ng * 64).reshape(ng, 64)
    fsum = _fetch(prob, "fsum", frames * 32).reshape(frames, 32)
    prob.close()
    assert np.isclose(c0, c1, rtol=1e-13)
    off, cam = np.asarray(sc["frame_offsets"]), np.asarray(sc["obs_cam"])
    gframe = np.concatenate([[f] * len(np.unique(cam[off[f]:off[f + 1]])) for f in range(frames)])
    gcam = np.concatenate([np.unique(cam[off[f]:off[f + 1]]) for f in range(frames)])
    idx7 = [0, 1, 2, 3, 4, 5, 12]
    tri = [(i, j) for i in range(7) for j in range(i + 1)]
    tri6 = [(i, j) for i in range(6) for j in range(i + 1)]
    for g in range(ng):
        tile = b0[g]
        scale = np.abs(tile).max()
        if not frozen[gcam[g]]:   # (a camera held constant: the tile form zeroes its rows, the record keeps G7 -- nobody reads it)
            g7 = np.array([tile[idx7[i], idx7[j]] for i, j in tri])
            assert np.abs(rec[g, :28] - g7).max() < 1e-11 * scale
            assert np.abs(rec[g, 28:].reshape(6, 6) - tile[0:6, 6:12]).max() < 1e-11 * scale
    for f in range(frames):
        sel = gframe == f
        if not sel.any():
            continue
        hff = b0[sel][:, 6:12, 6:12].sum(axis=0)
        gf = b0[sel][:, 6:12, 12].sum(axis=0)
        scale = np.abs(hff).max()
        assert np.abs(fsum[f, :21] - np.array([hff[i, j] for i, j in tri6])).max() < 1e-11 * scale
        assert np.abs(fsum[f, 21:27] - gf).max() < 1e-11 * max(scale, np.abs(gf).max())


@pytest.mark.parametrize("frame", [0, 1])
@pytest.mark.parametrize("cams,frames,pts", [(3, 30, 150), (4, 40, 30), (2, 300, 4)])
def test_rig_solve_matches_the_oracle_under_either_sweep(monkeypatch, frame, cams, frames, pts):
    monkeypatch.setenv("CC_RIG_SWEEP_FRAME", str(frame))
    if frame:
        monkeypatch.setenv("CC_RIG_PERSIST", "0")   # (small rigs run the lean persistent form by default: the frame form is the three-kernel path's)
    sc = po.rig_scenario(cams, frames, pts)
    g, o = _both(sc, cams)
    _assert_same(g, o)


@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("cams,frames,pts,mask,huber_a", [(3, 20, 300, 0, 0.0), (4, 60, 30, (1 << 8) | (1 << 6), 1.5)])
def test_rigk_solve_matches_the_oracle_under_either_sweep(monkeypatch, compact, cams, frames, pts, mask, huber_a):
    monkeypatch.setenv("CC_RIG_K_COMPACT", str(compact))
    k = rigk_case(cams, frames, pts)
    g, o = _both_k(k, const_mask=mask, huber_a=huber_a)
    _assert_same_k(g, o)


@pytest.mark.parametrize("waves", [1, 4])
@pytest.mark.parametrize("per_camera", [False, True])
def test_rigk_sweep_workgroup_sizes_give_the_same_solve(monkeypatch, waves, per_camera):
    """The sweep with intrinsics runs with one wave per (frame, camera) group when there are at least 1024 groups and with
    four otherwise; either must match the oracle on a shape the heuristic would give the other (ragged last chunks: 90 and
    300 observations per group)."""
    from tests.test_gpu_rigk import _assert_same_pc, _both_pc
    monkeypatch.setenv("CC_RIG_K_COMPACT", "0")   # (the tile form: k_rig_sweep_k2 has one workgroup shape)
    monkeypatch.setenv("CC_RIG_SWEEP_WG_WAVES", str(waves))
    for cams, frames, pts in [(3, 24, 90), (2, 12, 300)]:
        k = rigk_case(cams, frames, pts, per_camera=per_camera)
        if per_camera:
            g, o = _both_pc(k, huber_a=1.5)
            _assert_same_pc(g, o)
        else:
            g, o = _both_k(k, const_mask=1 << 5, huber_a=1.5)
            _assert_same_k(g, o)


@pytest.mark.parametrize("per_camera", [False, True])
@pytest.mark.parametrize("cams,frames,pts,mask,huber_a", [(3, 24, 90, 1 << 7, 2.0), (2, 12, 300, 0, 0.0), (4, 10, 131, 1 << 8, 1.5), (2, 30, 5, 0, 2.0)])
def test_compact_k_records_equal_the_tiles(monkeypatch, per_camera, cams, frames, pts, mask, huber_a):
    """Round 5: the sweep with intrinsics (k_rig_sweep_k2: plain FMAs on the lower triangle of the 16-column Gram, two waves
    per group that split the pairs) writes ONE record of 256 doubles per group -- direct sums, coupling columns T = G_cc M and
    H_fk = M^T H_ck, the group's share of the frame block and gradient -- where k_rig_sweep_adjk (matrix pipe,
    CC_RIG_K_COMPACT=0) writes three 16 x 16 tiles. Every record entry is an entry of those tiles: compared one by one,
    ragged groups (pts not a multiple of 128, fewer observations than a wave) and a constant intrinsic included."""
    k = rigk_case(cams, frames, pts, per_camera=per_camera)
    ng = _group_count(k, cams)
    out = []
    for compact in (0, 1):
        monkeypatch.setenv("CC_RIG_K_COMPACT", str(compact))
        prob = capi.RigProblem(cams, k["frame_offsets"], k["obs_cam"], k["obs_world"], k["obs_uv_pix"], k["world_xyz"], k["cam_frozen"],
                               huber_a=huber_a, with_intrinsics="per_camera" if per_camera else True)
        if per_camera:
            for c in range(cams):
                prob.set_camera_intrinsics(c, k["intr0"][c], mask)
        else:
            prob.set_intrinsics(k["intr0"], mask)
        prob.set_state(k["cam_q0"], k["cam_t0"], k["frame_q0"], k["frame_t0"])
        cost = prob.solve(capi.default_options(max_iterations=1))["initial_cost"]
        data = _fetch(prob, "gcomp", ng * 256).reshape(ng, 256) if compact else _fetch(prob, "gblocks", ng * 768).reshape(ng, 3, 16, 16)
        prob.close()
        out.append((cost, data))
    (c0, tiles), (c1, rec) = out
    assert np.isclose(c0, c1, rtol=1e-13)
    off, cam = np.asarray(k["frame_offsets"]), np.asarray(k["obs_cam"])
    gcam = np.concatenate([np.unique(cam[off[f]:off[f + 1]]) for f in range(frames)])
    tri6 = [(i, j) for i in range(6) for j in range(i + 1)]
    tri9 = [(i, j) for i in range(9) for j in range(i + 1)]
    worst = 0.0
    for g in range(ng):
        AA, AB, BB = tiles[g]
        scale = max(np.abs(AA).max(), np.abs(AB).max(), np.abs(BB).max())
        want = {}
        if not k["cam_frozen"][gcam[g]]:   # (a camera held constant: the tiles zero its rows, the record keeps the sums -- nobody reads them)
            want[(0, 21)] = [AA[i, j] for i, j in tri6]
            want[(21, 27)] = AA[0:6, 12]
            want[(27, 81)] = AB[0:6, 0:9].ravel()
            want[(136, 172)] = AA[0:6, 6:12].ravel()
        want[(81, 126)] = [BB[i, j] for i, j in tri9]
        want[(126, 135)] = AB[12, 0:9]
        want[(135, 136)] = [AA[12, 12]]
        want[(172, 226)] = AB[6:12, 0:9].ravel()
        want[(226, 247)] = [AA[6 + i, 6 + j] for i, j in tri6]
        want[(247, 253)] = AA[6:12, 12]
        for (a, b), w in want.items():
            # per kind of entry against the largest entry of that kind in the group (the tiles mix pixel^2 and O(1) numbers)
            sc = max(np.abs(np.asarray(w)).max(), 1e-300)
            worst = max(worst, np.abs(rec[g, a:b] - np.asarray(w)).max() / sc)
        assert np.array_equal(rec[g, 253:], np.zeros(3))
        assert scale > 0
    assert worst < 1e-11, worst
    if mask:
        q = int(np.log2(mask))   # the intrinsic held constant has no column
        assert np.abs(rec[:, 27:81].reshape(ng, 6, 9)[:, :, q]).max() == 0 and np.abs(rec[:, 172:226].reshape(ng, 6, 9)[:, :, q]).max() == 0
