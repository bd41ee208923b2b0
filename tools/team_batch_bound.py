"""What register-resident multi-CU members could deliver on BATCHES of large pairs (round-4 verdict item 1), measured
with the kernels that exist: the team kernel (one patch per lane, grid in VGPRs for the level, members of 4 patch waves
exchanging 30 doubles per iteration through tagged words) on as many pairs as fit HALF the chip at once (its admission
rule), against the workspace kernels on the same pairs and on the full batch.

A persistent team kernel over a batch would run rounds of exactly such launches, two of them side by side on the two
halves of the chip: its rate is bounded by 2 x pairs_per_half / t_team (no round is shorter than its slowest pair).

    python tools/team_batch_bound.py            (MI355X; prints one table per shape)
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


def time_launches(ctx, d, desc, cs, prm, stream, n=12):
    ev = []
    with torch.cuda.stream(stream):
        for k in range(n + 3):
            d["T_cur_w"].copy_(d["T_seed"])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(desc), C.byref(cs), C.byref(prm), stream.cuda_stream))
            b.record(stream)
            if k >= 3:
                ev.append((a, b))
    stream.synchronize()
    ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, stream.cuda_stream))
    t = [a.elapsed_time(b) for a, b in ev]
    return float(np.mean(t)), float(np.min(t))


def main():
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, diag=True)   # the diagnostic library (dsdtm_debug_* / switches)
    stream = torch.cuda.Stream(device=dev)
    setopt = ctx.lib.dsdtm_debug_set_option
    for name, W, H, n_full, N in (("config 3 shape", 640, 480, 1024, 1000), ("config 5 shape", 1280, 960, 256, 2000)):
        cam = synth.Camera.tum(W, H)
        cs = capi.camera_struct(cam)
        prm = capi.AlignParams(4, 0, 10, 15)
        k = (N + 255) // 256
        n_half = 128 // k
        d = bench.build_batch(torch, dev, ctx, cam, n_full, W, H, 4, N, seed=0xC0DE + N, stream=stream)
        desc = d["desc"]
        sub = capi.BatchDesc.from_buffer_copy(bytes(desc))
        sub.n_pairs = n_half
        print(f"== {name}: {W}x{H}, {N} patches; team size {k}, {n_half} pairs fill half the chip", flush=True)
        setopt(b"no_team", 0)
        t_team = time_launches(ctx, d, sub, cs, prm, stream)
        Tt = d["T_cur_w"][:n_half].clone()
        setopt(b"no_team", 1)
        t_ws_half = time_launches(ctx, d, sub, cs, prm, stream)
        same = bool(torch.allclose(Tt, d["T_cur_w"][:n_half], atol=1e-9, rtol=0))
        t_ws_full = time_launches(ctx, d, desc, cs, prm, stream)
        setopt(b"no_team", 0)
        print(f"   team kernel, {n_half} pairs ({n_half * k} workgroups):      {t_team[0]:.4f} ms (min {t_team[1]:.4f})"
              f"  -> bound for a persistent team batch kernel: {2 * n_half / t_team[0] / 1e3:.3f} M alignments/s")
        print(f"   one-CU kernels, the same {n_half} pairs:                 {t_ws_half[0]:.4f} ms (min {t_ws_half[1]:.4f}); poses equal to 1e-9: {same}")
        print(f"   one-CU kernels, {n_full} pairs (the bench's secondary): {t_ws_full[0]:.4f} ms (min {t_ws_full[1]:.4f})"
              f"  = {n_full / t_ws_full[0] / 1e3:.3f} M alignments/s", flush=True)
        del d
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
