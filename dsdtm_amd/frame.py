"""Host-side data model the hot path reads, mirroring the reference's types just far enough
for the path: Config keys, Camera, Frame (pyramid + pose + features).

Struct-of-arrays instead of `std::vector<Feature*>`: the C ABI takes the feature columns
directly (include/dsdtm_amd.h), so the adapter does no per-feature marshalling.
"""
from __future__ import annotations

import numpy as np

from .synth import Camera  # float32 intrinsics, reference include/Camera.h:138-142


class Config:
    """Stand-in for the reference's cv::FileStorage singleton (include/Config.h:28-31).
    Only the keys the path reads exist: defaults from Config/default.yaml."""
    _values = {
        "Camera.Min_fts": 15,          # src/Sprase_ImageAlign.cpp:14
        "Camera.Max_tkfts": 200,       # src/Feature_alignment.cpp:24
        "Camera.MaxPyraLevels": 5,     # :25, src/Tracking.cpp:20
        "Camera.MinPyraLevels": 0,
        "Camera.CellSize": 25,         # :28
        "Optimization.MaxIter": 8,     # src/Tracking.cpp:24
        "Camera.Max_fts": 200,         # src/Feature_detection.cpp:14 (Config/EuRoc.yaml:27)
        "Camera.Min_dist": 30,         # src/Frame.cpp:52 (Config/EuRoc.yaml:28)
        "Optimization.LocalBAthreshhold": 2.0,   # src/Optimizer.cpp:22 (Config/default.yaml:94)
    }

    @classmethod
    def Get(cls, key):
        return cls._values[key]

    @classmethod
    def Set(cls, key, value):
        cls._values[key] = value

    @classmethod
    def setParameterFile(cls, path):
        import yaml
        with open(path) as f:
            text = f.read()
        if text.startswith("%YAML"):
            text = "\n".join(text.split("\n")[1:])
        data = yaml.safe_load(text.replace("---", "", 1)) or {}
        for k, v in data.items():
            cls._values[k] = v


class Frame:
    """What Sprase_ImgAlign / Feature_Alignment read from DSDTM::Frame (include/Frame.h):
    mvImg_Pyr, mT_c2w (world->camera, [R|t] 3x4), mOw, and the feature columns
    Feature::{mpx, mlevel, mNormal, mbInitial, Mpt->Get_Pose()} (include/Feature.h:16-36)."""

    def __init__(self, camera: Camera, img_pyr, T_c2w=None):
        self.mCamera = camera
        self.mvImg_Pyr = [np.ascontiguousarray(l, dtype=np.uint8) for l in img_pyr]
        self.px = np.zeros((0, 2), np.float32)
        self.level = np.zeros((0,), np.int32)
        self.bearing = np.zeros((0, 3), np.float64)
        self.p_world = np.zeros((0, 3), np.float64)
        self.initial = np.zeros((0,), np.uint8)
        self.Set_Pose(np.eye(4)[:3] if T_c2w is None else T_c2w)

    # Frame::Set_Pose / Get_Pose / Get_CameraCnt (src/Frame.cpp:167-174, include/Frame.h:50-54)
    def Set_Pose(self, T):
        self.mT_c2w = np.array(T, dtype=np.float64).reshape(-1, 4)[:3].copy()
        R, t = self.mT_c2w[:, :3], self.mT_c2w[:, 3]
        self.mOw = -R.T @ t

    def Get_Pose(self):
        return self.mT_c2w

    def Get_CameraCnt(self):
        return self.mOw

    def set_features(self, px, bearing, p_world, initial, level=None):
        n = len(px)
        self.px = np.ascontiguousarray(px, np.float32).reshape(n, 2)
        self.bearing = np.ascontiguousarray(bearing, np.float64).reshape(n, 3)
        self.p_world = np.ascontiguousarray(p_world, np.float64).reshape(n, 3)
        self.initial = np.ascontiguousarray(initial, np.uint8).reshape(n)
        self.level = np.zeros(n, np.int32) if level is None else np.ascontiguousarray(level, np.int32)

    @property
    def n_features(self):
        return len(self.px)

    # Frame::World2Pixel (src/Frame.cpp:318-323)
    def World2Pixel(self, P):
        p = self.mT_c2w[:, :3] @ np.asarray(P, np.float64) + self.mT_c2w[:, 3]
        c = self.mCamera
        return np.array([c.fx * p[0] / p[2] + c.cx, c.fy * p[1] / p[2] + c.cy])


def frames_from_scene(scene):
    """(cur, ref) Frames of a synth.AlignScene, seeded as Tracking does (src/Tracking.cpp:201)."""
    ref = Frame(scene.cam, scene.ref_pyr, scene.T_ref_w)
    ref.set_features(scene.px, scene.bearing, scene.p_world, scene.initial)
    cur = Frame(scene.cam, scene.cur_pyr, scene.T_cur_w_seed)
    return cur, ref
