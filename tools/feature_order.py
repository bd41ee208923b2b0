"""Does the ORDER of a pair's features matter to the batched large-pair kernels? The same 1024 x 1000-patch (640x480)
and 256 x 2000-patch (1280x960) batches with the feature columns permuted: as generated (uniformly random positions),
sorted by 128-px strip then row, by row, and along a Morton curve. Lanes of a wave then gather from neighbouring rows /
the same cache lines. Results must not change beyond summation order (checked against the unsorted run).

    python tools/feature_order.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dsdtm_amd import capi, synth  # noqa: E402
from tools.team_batch_bound import time_launches  # noqa: E402


def morton(x, y):
    def spread(v):
        v = v.astype(np.uint64) & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)


def main():
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream(device=dev)
    for name, W, H, n_pairs, N in (("config 3 shape", 640, 480, 1024, 1000), ("config 5 shape", 1280, 960, 256, 2000)):
        cam = synth.Camera.tum(W, H)
        cs = capi.camera_struct(cam)
        prm = capi.AlignParams(4, 0, 10, 15)
        d = bench.build_batch(torch, dev, ctx, cam, n_pairs, W, H, 4, N, seed=0xC0DE + N, stream=stream)
        desc = d["desc"]
        px = d["px"].cpu().numpy()
        cols0 = {k: d[k].clone() for k in ("px", "bearing", "p_world", "initial")}
        orders = {
            "as generated (random)": None,
            "128-px strip, then row": (px[:, :, 0] // 128).astype(np.int64) * 100000 + px[:, :, 1].astype(np.int64),
            "row": px[:, :, 1].astype(np.int64) * 4096 + px[:, :, 0].astype(np.int64),
            "16-row band, then column": (px[:, :, 1] // 16).astype(np.int64) * 4096 + px[:, :, 0].astype(np.int64),
            "Morton": morton(px[:, :, 0].astype(np.int64), px[:, :, 1].astype(np.int64)).astype(np.int64),
        }
        print(f"== {name}: {n_pairs} pairs x {N} patches, {W}x{H}", flush=True)
        T0 = None
        for oname, key in orders.items():
            if key is not None:
                idx = torch.from_numpy(np.argsort(key, axis=1, kind="stable")).to(dev)
                for k, v in cols0.items():
                    ix = idx if v.dim() == 2 else idx[:, :, None].expand(-1, -1, v.shape[2])
                    d[k].copy_(torch.gather(v, 1, ix))
            t = time_launches(ctx, d, desc, cs, prm, stream)
            if T0 is None:
                T0 = d["T_cur_w"].clone()
            dmax = float((d["T_cur_w"] - T0).abs().max().item())
            print(f"   {oname:28s} {t[0]:.4f} ms (min {t[1]:.4f}) = {n_pairs / t[0] / 1e3:.3f} M alignments/s; max |pose - unsorted| {dmax:.1e}", flush=True)
        del d
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
