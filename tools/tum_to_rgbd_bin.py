"""A dataset in the TUM RGB-D benchmark's layout -> the container dsdtm_amd/host/example_rgbd.cpp reads (BASELINE config 1
through the C++ driver: the first frame is the reference, every later frame is aligned against it).

    python tools/tum_to_rgbd_bin.py <dataset_dir> <sequence.bin> [--frames N] [--levels 4] [--threshold 20] [--max-fts 400]
                                    [--fx .. --fy .. --cx .. --cy .. --f ..]      (default: TUM fr1 intrinsics scaled to the image)
    dsdtm_amd/host/example_rgbd <sequence.bin> <features_out.bin>

Per frame: the grayscale image (cv::imread(.., GRAYSCALE) of rgb/*.png) and its ground-truth pose (world -> camera, from
groundtruth.txt; identity where the file or a close timestamp is missing); for the first frame also the depth image in metres
(depth/*.png / Camera.depth_scale; 0 = no measurement — the driver drops features without depth). No GPU needed."""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsdtm_amd import synth, tum  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("out")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--threshold", type=int, default=20)
    ap.add_argument("--max-fts", type=int, default=400)
    ap.add_argument("--depth-scale", type=float, default=5000.0)
    for k in ("fx", "fy", "cx", "cy", "f"):
        ap.add_argument("--" + k, type=float, default=None)
    a = ap.parse_args()
    seq = tum.TumSequence(a.dataset, depth_scale=a.depth_scale)
    n = len(seq) if a.frames <= 0 else min(a.frames, len(seq))
    t0, gray0, depth0, _ = seq.frame(0)
    Hh, W = gray0.shape
    cam = synth.Camera.tum(W, Hh)
    vals = [getattr(a, k) if getattr(a, k) is not None else getattr(cam, k) for k in ("fx", "fy", "cx", "cy", "f")]
    with open(a.out, "wb") as f:
        f.write(struct.pack("<6i", W, Hh, a.levels, n, a.threshold, a.max_fts))
        f.write(struct.pack("<5f", *vals))
        for k in range(n):
            t, gray, depth, T_wc = seq.frame(k)
            T_cw = np.linalg.inv(T_wc)[:3] if T_wc is not None else np.eye(4)[:3]
            f.write(np.ascontiguousarray(gray, np.uint8).tobytes() + np.ascontiguousarray(T_cw, "<f8").tobytes())
            if k == 0:
                f.write(np.ascontiguousarray(depth, "<f4").tobytes())
    print(f"wrote {a.out}: {n} frames of {W}x{Hh}, {a.levels} levels")


if __name__ == "__main__":
    main()
