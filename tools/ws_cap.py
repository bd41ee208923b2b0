"""Where a launch of the large-pair kernels spends its time: the config-3 / config-5 batches at iteration caps 1 and 10
(cap 1 = four level starts with their first pass, H pass and solve; the difference = the later iterations). One line
per shape and cap; run once per library build in the same gpurun call (tools/ab_lib.sh) for same-box comparisons.

    python tools/ws_cap.py [label]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dsdtm_amd import capi, synth  # noqa: E402
from tools.team_batch_bound import time_launches  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "build"
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream(device=dev)
    for name, W, H, n_pairs, N in (("1024 x 1000 @640x480", 640, 480, 1024, 1000), ("256 x 2000 @1280x960", 1280, 960, 256, 2000)):
        cam = synth.Camera.tum(W, H)
        cs = capi.camera_struct(cam)
        d = bench.build_batch(torch, dev, ctx, cam, n_pairs, W, H, 4, N, seed=0xC0DE + N, stream=stream)
        for cap in (1, 10):
            prm = capi.AlignParams(4, 0, cap, 15)
            t = time_launches(ctx, d, d["desc"], cs, prm, stream)
            print(f"{label}: {name} cap {cap:2d}: {t[0]:.4f} ms (min {t[1]:.4f}) = {n_pairs / t[0] / 1e3:.3f} M alignments/s", flush=True)
        del d
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
