"""Pure-Python restatement of the ExtrinsicsCalibrator id bookkeeping and JSON wire format
(/root/reference/src/extrinsics_calibrator.cpp:9-49, 268-452). TEST INFRASTRUCTURE ONLY.

size_t arithmetic is emulated modulo 2**64 so that the reference's wrap-around in
RemoveObservationFrame (ids of later frames minus the removed frame's point count) is reproduced.
"""
import json
import math

import numpy as np

_M = 1 << 64


class Bookkeeping:
    def __init__(self):
        self.camera_T_rigs = []          # 4x4 float32
        self.frozen = set()
        self.frames = []                 # dict(rig_T_world, world_points[], observations[])
        self.world_point_infos = []      # (frame_id, world_point_idx)

    # extrinsics_calibrator.cpp:9-17
    def add_camera(self, T, freeze=False):
        self.camera_T_rigs.append(np.array(T, dtype=np.float32))
        i = len(self.camera_T_rigs) - 1
        if freeze:
            self.frozen.add(i)
        return i

    # :19-23
    def add_frame(self, T):
        self.frames.append(dict(rig_T_world=np.array(T, dtype=np.float32), world_points=[], observations=[]))
        return len(self.frames) - 1

    # :25-34
    def add_world_point(self, frame_id, p):
        self.world_point_infos.append([frame_id, len(self.frames[frame_id]["world_points"])])
        self.frames[frame_id]["world_points"].append(np.array(p, dtype=np.float32))
        return len(self.world_point_infos) - 1

    # :36-49
    def add_observation(self, camera_id, world_point_id, uv):
        frame_id, idx = self.world_point_infos[world_point_id]
        self.frames[frame_id]["observations"].append(
            dict(camera_id=camera_id, world_point_idx=idx, world_point_id=world_point_id,
                 image_point=np.array(uv, dtype=np.float32), cost=float("nan")))

    # :415-442
    def remove_frame(self, k):
        n = len(self.frames[k]["world_points"])
        del self.frames[k]
        for f in self.frames[k:]:
            for o in f["observations"]:
                o["world_point_id"] = (o["world_point_id"] - n) % _M
        self.world_point_infos = [w for w in self.world_point_infos if w[0] != k]
        for w in self.world_point_infos:
            if w[0] >= k:
                w[0] -= 1

    # :444-452
    def remove_frames(self, ids):
        for k in sorted(ids, reverse=True):
            self.remove_frame(k)

    # :268-346 -- nlohmann::json dump: keys sorted, compact, NaN -> null, 16 floats column-major
    def to_json_obj(self):
        def tr(T):
            return [float(v) for v in np.asarray(T, dtype=np.float32).T.reshape(-1)]
        return {
            "camera_T_rigs": [{"camera_T_rig": tr(T), "frozen": i in self.frozen} for i, T in enumerate(self.camera_T_rigs)],
            "observation_frames": [
                {"observations": [{"camera_id": o["camera_id"], "cost": None if math.isnan(o["cost"]) else o["cost"],
                                   "image_point": [float(v) for v in o["image_point"]], "world_point_id": o["world_point_id"]}
                                  for o in f["observations"]],
                 "rig_T_world": tr(f["rig_T_world"])} for f in self.frames],
            "world_points": [{"frame_id": w[0], "world_point": [float(v) for v in self.frames[w[0]]["world_points"][w[1]]]}
                             for w in self.world_point_infos],
        }

    def dumps(self):
        return json.dumps(self.to_json_obj(), separators=(",", ":"), sort_keys=True)
