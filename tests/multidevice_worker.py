"""Body of tests/test_gpu_multidevice.py, run in a process of its own so that GPU_MAX_HW_QUEUES can be raised
before the HIP runtime starts: the test box has ONE GPU, so the "devices" of the one-host-thread multi-device
drivers (cc_intrinsics_optimize_multi / cc_rig_optimize_multi, Calibrator::SetDevices) are device 0 several
times; their kernels wait for each other's mailbox posts and must therefore sit on different hardware queues."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "camera_calibrator_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from camera_calibrator_amd import capi  # noqa: E402


def same_solve(a, b, what):
    sa, sb = a[-1], b[-1]
    assert sa["iterations"] == sb["iterations"] and sa["termination"] == sb["termination"], (what, sa["iterations"], sb["iterations"])
    assert [l["accepted"] for l in sa["log"]] == [l["accepted"] for l in sb["log"]], what
    assert np.allclose([l["cost"] for l in sa["log"]], [l["cost"] for l in sb["log"]], rtol=1e-9), what
    assert np.isclose(sa["final_cost"], sb["final_cost"], rtol=1e-10), what


def intrinsics():
    off, uv, xyz = capi.make_intrinsics_problem(61, [40 + (7 * f) % 90 for f in range(61)])      # ragged frames
    K0, q0, t0 = capi.zhang_init(off, uv, xyz)
    intr0 = np.array([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    q0, t0 = q0.astype(np.float64), t0.astype(np.float64)
    one = capi.intrinsics_optimize(off, uv, xyz, intr0, q0, t0, const_mask=1 << 8)
    for devs in ([0, 0], [0, 0, 0, 0], [0, 0, 0]):
        many = capi.intrinsics_optimize(off, uv, xyz, intr0, q0, t0, const_mask=1 << 8, devices=devs)
        same_solve(one, many, f"intrinsics {devs}")
        assert np.allclose(many[0][:4], one[0][:4], rtol=1e-9) and np.allclose(many[0][4:], one[0][4:], atol=1e-9)
        assert np.abs(many[1] - one[1]).max() < 1e-9 and np.abs(many[2] - one[2]).max() < 1e-9 and many[0][8] == 0.0
    # more devices than frames: extra devices stay idle
    off3, uv3, xyz3 = capi.make_intrinsics_problem(3, 50)
    K3, q3, t3 = capi.zhang_init(off3, uv3, xyz3)
    i3 = np.array([K3[0, 0], K3[1, 1], K3[0, 2], K3[1, 2], 0, 0, 0, 0, 0], dtype=np.float64)
    a = capi.intrinsics_optimize(off3, uv3, xyz3, i3, q3.astype(np.float64), t3.astype(np.float64), const_mask=0b111110000)
    b = capi.intrinsics_optimize(off3, uv3, xyz3, i3, q3.astype(np.float64), t3.astype(np.float64), const_mask=0b111110000, devices=[0] * 8)
    same_solve(a, b, "3 frames on 8 devices")
    print("intrinsics multi-device ok")


def calibrator_class():
    import pycalibrator as pc
    off, uv, xyz = capi.make_intrinsics_problem(40, 100)
    img = [uv[off[f]:off[f + 1]] for f in range(40)]
    world = [xyz[off[f]:off[f + 1]] for f in range(40)]
    ref = pc.Calibrator(1600, 1000)
    ref.Estimate(img, world)
    for devs in ([0, 0], [0, 0, 0, 0]):
        c = pc.Calibrator(1600, 1000)
        c.SetDevices(devs)
        c.Estimate(img, world)
        assert c.LastStatus() == 0 and c.LastIterations() == ref.LastIterations()
        assert np.isclose(c.LastFinalCost(), ref.LastFinalCost(), rtol=1e-10)
        got = np.concatenate([np.asarray(c.GetK()).ravel(), np.asarray(c.GetDistortion()).ravel()]).astype(np.float32)
        want = np.concatenate([np.asarray(ref.GetK()).ravel(), np.asarray(ref.GetDistortion()).ravel()]).astype(np.float32)
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1, (devs, got, want)
    print("Calibrator.SetDevices ok")


def rig():
    sc = capi.rig_scenario(3, 41, 12)
    cq, ct = capi.affine_to_qt(sc["cam_T"])
    fq, ft = capi.affine_to_qt(sc["frame_T"])
    # camera 2 is seen by the last frames only: the first shards do not observe it, the column layout must still agree
    keep = (sc["obs_cam"] != 2) | (np.repeat(np.arange(41), 36) >= 30)
    counts = [int(keep[sc["frame_offsets"][f]:sc["frame_offsets"][f + 1]].sum()) for f in range(41)]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    args = (3, offs, sc["obs_cam"][keep], sc["obs_world"][keep], sc["obs_uv"][keep], sc["world_xyz"], cq, ct, sc["cam_frozen"], fq, ft)
    o = capi.default_options(max_iterations=1000)
    one = capi.rig_optimize(*args, options=o)
    for devs in ([0, 0], [0, 0, 0], [0, 0, 0, 0]):
        many = capi.rig_optimize(*args, options=o, devices=devs)
        same_solve(one, many, f"rig {devs}")
        for k in range(4):
            assert np.abs(many[k] - one[k]).max() < 1e-9, (devs, k)
        assert np.allclose(many[4], one[4], rtol=1e-8, atol=1e-16)
    print("rig multi-device ok")


def rig_class():
    import pycalibrator as pc

    def build():
        e = pc.ExtrinsicsCalibrator()
        e.SetVerbose(False)
        sc = capi.rig_scenario(2, 30, 4)
        for c in range(2):
            e.AddCameraTRig(sc["cam_T"][c].reshape(4, 4).T, c == 0)
        k = 0
        for f in range(30):
            e.AddObservationFrame(sc["frame_T"][f].reshape(4, 4).T)
            for p in range(4):
                wp = e.AddWorldPoint(f, sc["world_xyz"][f * 4 + p])
                for c in range(2):
                    e.AddObservation(c, wp, sc["obs_uv"][k])
                    k += 1
        return e
    ref = build()
    ref.Optimize()
    e = build()
    e.SetDevices([0, 0, 0])
    e.Optimize()
    assert e.LastStatus() == 0 and e.LastIterations() == ref.LastIterations()
    assert np.isclose(e.LastFinalCost(), ref.LastFinalCost(), rtol=1e-10)
    assert np.abs(np.asarray(e.GetCameraTRig(1)) - np.asarray(ref.GetCameraTRig(1))).max() <= 2e-7
    print("ExtrinsicsCalibrator.SetDevices ok")


if __name__ == "__main__":
    which = sys.argv[1:] or ["intrinsics", "calibrator_class", "rig", "rig_class"]
    for w in which:
        globals()[w]()
