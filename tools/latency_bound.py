"""Where the device time of ONE Run goes, and what spreading it over more compute units costs (round-4 verdict item 3).

For BASELINE configs 2 / 3 / 5 (one pair each), HIP events on the launch stream around dsdtm_sparse_align_batch_device:
  * iteration caps 1 and 10: cap 1 = the level starts with their first pass + solve; the difference / the extra executed
    iterations = one later iteration (pass -> reduce -> [exchange] -> solve -> publish);
  * config 2 on ONE compute unit (the register kernel, 5 + 1 waves) and as a TEAM of two (team_min = 1: the same per-lane
    work, 150 patches per member, partials exchanged through tagged words in L2 every iteration): the difference is what
    the cross-CU exchange costs against what the halved level-start gathers return;
  * an empty stream operation between the same events: the floor of the measurement itself.

    python tools/latency_bound.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench_tracking as BT  # noqa: E402
from dsdtm_amd import capi, synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, diag=True)   # the diagnostic library (dsdtm_debug_* / switches)
    stream = torch.cuda.Stream(device=dev)
    setopt = ctx.lib.dsdtm_debug_set_option
    z = torch.zeros(1, dtype=torch.int32, device=dev)
    floor, _ = BT._median_device(torch, stream, lambda: z.zero_(), n=100)
    print(f"event pair around one 4-byte memset on the stream: {floor * 1e3:.1f} us (the floor of every figure below)")
    for name, kw, L in (("config 2: 640x480, 300 patches", dict(), 4),
                        ("config 3: 640x480, 1000 patches", dict(n_patches=1000, cam=synth.Camera.tum(640, 480, synth.TUM_FR3)), 4),
                        ("config 5: 1280x960, 2000 patches", dict(width=1280, height=960, n_patches=2000, margin=60), 4)):
        sc = synth.make_scene(**kw)
        t, b, _ = BT._device_pair(torch, dev, capi, sc, L)
        cs = capi.camera_struct(sc.cam)
        st = torch.zeros((1, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        b.stats = st.data_ptr()
        print(f"== {name}", flush=True)
        variants = [("default dispatch", {})]
        if len(sc.px) <= 448:
            variants.append(("team of 2 compute units (team_min = 1)", {b"team_min": 1}))
        else:
            variants.append(("one-CU kernels (no_team = 1)", {b"no_team": 1}))
        for vname, opts in variants:
            for k, v in opts.items():
                setopt(k, v)
            res = {}
            for cap in (1, 10):
                ap = capi.AlignParams(L, 0, cap, 15)
                ms, mn = BT._median_device(
                    torch, stream,
                    lambda: ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cs), C.byref(ap), stream.cuda_stream)),
                    n=100, before=lambda: t["Tc"].copy_(t["seed"], non_blocking=True))
                ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, stream.cuda_stream))
                its = int(np.frombuffer(st.cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)["iters"][0][:L].sum())
                res[cap] = (ms, its)
            (m1, i1), (m10, i10) = res[1], res[10]
            per_it = (m10 - m1) / max(1, i10 - i1)
            print(f"   {vname:42s} cap 1: {m1 * 1e3:6.1f} us ({i1} iterations)   cap 10: {m10 * 1e3:6.1f} us ({i10} iterations)"
                  f"   -> {per_it * 1e3:5.2f} us per later iteration, {(m1 - floor) / L * 1e3:5.1f} us per level start incl. its first iteration", flush=True)
            for k in opts:
                setopt(k, 449 if k == b"team_min" else 0)
        del t


if __name__ == "__main__":
    main()
