"""BASELINE configs 1 / 3 on a dataset in the TUM RGB-D benchmark's layout (associations.txt, rgb/, depth/, optionally
groundtruth.txt) — the data the reference's own drivers read (Test/test_Tracking.cpp:56-82; Test/test_SpraseImg_alignment.cpp
:85-168 for the frame-to-reference flow followed here).

For every frame k > 0: features of the reference frame (the product's Feature_detector on its pyramid), their depth from the
depth image (Frame::Get_FeatureDetph), 3-D points in the world, then Sprase_ImgAlign::Run(cur, ref) seeded with the previous
pose. The reference frame is re-anchored every `--keyframe-every` frames (test_SpraseImg_alignment keeps the first frame;
Tracking aligns against the last frame, src/Tracking.cpp:199-217: --keyframe-every 1). Writes CameraTrajectory.txt in the
benchmark's format; with groundtruth.txt present prints the translational error per frame against it.

    python tools/run_tum.py <dataset_dir> [--config Config/default.yaml] [--frames N] [--keyframe-every K] [--out traj.txt]

Needs an MI355X (the image work runs through libdsdtm_amd.so; there is no CPU path). No dataset ships with this repository:
tests/test_tum_format.py builds a miniature one in the same file formats and runs this driver on it.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsdtm_amd import capi, synth, tum  # noqa: E402
from dsdtm_amd.feature_detection import Feature_detector  # noqa: E402
from dsdtm_amd.frame import Config, Frame  # noqa: E402
from dsdtm_amd.sparse_align import Sprase_ImgAlign  # noqa: E402


def reference_frame(cam, gray, depth, T_c2w, levels, detector, threshold):
    """A frame with the detector's corners, bearings, and map points from the depth image (src/Tracking.cpp:98-135 in
    spirit: features without a valid depth carry no map point — mbInitial false — and are skipped by Run, :86)."""
    fr = Frame(cam, synth.build_pyramid(gray, levels), T_c2w)
    detector.detect(fr, threshold)
    px = fr.px
    bearing = synth.bearing_from_px(cam, px)
    R, t = fr.Get_Pose()[:, :3], fr.Get_Pose()[:, 3]
    p_world = np.zeros((len(px), 3))
    initial = np.zeros(len(px), np.uint8)
    for i in range(len(px)):
        z = tum.get_feature_depth(depth, px[i])
        if z <= 0:
            continue
        Xc = bearing[i] * (z / bearing[i, 2])              # z-depth -> point on the viewing ray
        p_world[i] = R.T @ (Xc - t)
        initial[i] = 1
    fr.set_features(px, bearing, p_world, initial, fr.level)
    return fr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("--config", default=None, help="a reference-style YAML (Config/default.yaml keys); default: TUM fr1 intrinsics")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--keyframe-every", type=int, default=1)
    ap.add_argument("--out", default="CameraTrajectory.txt")
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    if a.config:
        Config.setParameterFile(a.config)
    seq = tum.TumSequence(a.dataset, depth_scale=float(Config._values.get("Camera.depth_scale", 5000)))
    t0, gray0, depth0, T_wc0 = seq.frame(0)
    Hh, W = gray0.shape
    if a.config:
        g = Config._values
        cam = synth.Camera(g["Camera.fx"], g["Camera.fy"], g["Camera.cx"], g["Camera.cy"], g.get("Camera.f", g["Camera.fx"]), W, Hh)
    else:
        cam = synth.Camera.tum(W, Hh)
    ctx = capi.default_context(0)
    Config.Set("Camera.MaxPyraLevels", a.levels)
    det = Feature_detector(W, Hh, ctx=ctx)
    al = Sprase_ImgAlign(a.levels, 0, a.iters, ctx=ctx)
    T0 = np.linalg.inv(T_wc0)[:3] if T_wc0 is not None else np.eye(4)[:3]     # world -> camera
    ref = reference_frame(cam, gray0, depth0, T0, a.levels, det, 20.0)
    print(f"frame 0: {ref.n_features} features, {int(ref.initial.sum())} with depth")
    n = len(seq) if a.frames <= 0 else min(a.frames, len(seq))
    stamps, poses = [t0], [T0.copy()]
    T_prev = T0.copy()
    for k in range(1, n):
        t, gray, depth, T_wc = seq.frame(k)
        cur = Frame(cam, synth.build_pyramid(gray, a.levels), T_prev)
        tracked = al.Run(cur, ref)
        T_prev = cur.Get_Pose().copy()
        stamps.append(t); poses.append(T_prev)
        msg = f"frame {k}: tracked {tracked} iterations {al.last_stats['iters'][:a.levels]}"
        if T_wc is not None:
            Cw = -T_prev[:, :3].T @ T_prev[:, 3]
            msg += f" translation_error {np.linalg.norm(Cw - T_wc[:3, 3]):.5f} m"
        print(msg, flush=True)
        if k % a.keyframe_every == 0:
            ref = reference_frame(cam, gray, depth, T_prev, a.levels, det, 20.0)
    tum.write_trajectory(a.out, stamps, poses)
    print(f"wrote {a.out} ({len(poses)} poses)")


if __name__ == "__main__":
    main()
