"""Soak of dsdtm_sparse_align_batch_streamed against dsdtm_sparse_align_batch_sharded: random pair counts, feature counts
(incl. ragged per-pair counts and the large-pair kernels), chunk sizes, context counts, chained / separate frames, padded host
rows. Every case must be bit-identical between the two entries (same frames, pyramids built on the host by the numpy pyrDown
for the sharded entry, on the device for the streamed one).
    python tools/soak_streamed.py [cases=40]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsdtm_amd import capi, synth  # noqa: E402
from tests.test_sharded_gpu import _host_batch, _stream_desc  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(2024)
    lib = capi.load()
    pool = [capi.Context(0) for _ in range(4)]
    bad = 0
    for case in range(n_cases):
        W, Hh = [(320, 240), (160, 120), (328, 248)][rng.integers(0, 3)]
        L = int(rng.integers(2, 4))
        N = int(rng.choice([40, 150, 300, 460, 720, 1100]))
        chained = bool(rng.integers(0, 2))
        P = int(rng.integers(1, 14))
        G = int(rng.integers(1, 5))
        chunk = int(rng.choice([1, 2, 3, 5, 8, 128]))
        padded = bool(rng.integers(0, 2))
        ragged = bool(rng.integers(0, 2))
        margin = 12
        if chained:
            scenes = synth.make_sequence(n_frames=P + 1, width=W, height=Hh, levels=L, n_patches=N, seed=int(rng.integers(1, 1 << 30)), margin=margin)
        else:
            base = [synth.make_scene(width=W, height=Hh, levels=L, n_patches=N, seed=int(rng.integers(1, 1 << 30)), margin=margin) for _ in range(min(P, 3))]
            scenes = [base[i % len(base)] for i in range(P)]
        cam = capi.camera_struct(scenes[0].cam)
        prm = capi.AlignParams(L, 0, int(rng.integers(1, 11)), 15)
        a0, b0, _ = _host_batch(scenes, L, W, Hh)
        nf = None
        if ragged:
            nf = rng.integers(5, N + 1, P).astype(np.int32)
            b0.n_features = nf.ctypes.data
        one = (C.c_void_p * 1)(pool[0].handle)
        rc = lib.dsdtm_sparse_align_batch_sharded(one, 1, C.byref(b0), C.byref(cam), C.byref(prm))
        assert rc == 0, pool[0].lib.dsdtm_last_error(pool[0].handle)
        RS = W + (int(rng.integers(1, 40)) if padded else 0)
        if chained:
            imgs = [sc.ref_pyr[0] for sc in scenes] + [scenes[-1].cur_pyr[0]]
            fr = np.full((P + 1, Hh, RS), 3, np.uint8)
            for i, im in enumerate(imgs):
                fr[i, :, :W] = im
            fc = None
        else:
            fr = np.full((P, Hh, RS), 3, np.uint8); fc = np.full((P, Hh, RS), 5, np.uint8)
            for i, sc in enumerate(scenes):
                fr[i, :, :W] = sc.ref_pyr[0]; fc[i, :, :W] = sc.cur_pyr[0]
        a, _, _ = _host_batch(scenes, L, W, Hh)
        s = _stream_desc(a, fr, fc, P, N, L, W, Hh, row_stride=RS, image_pitch=RS * Hh)
        if ragged:
            s.n_features = nf.ctypes.data
        arr = (C.c_void_p * G)(*[c.handle for c in pool[:G]])
        rc = lib.dsdtm_sparse_align_batch_streamed(arr, G, C.byref(s), chunk, C.byref(cam), C.byref(prm))
        assert rc == 0, [c.lib.dsdtm_last_error(c.handle) for c in pool[:G]]
        same = (np.array_equal(a["Tc"], a0["Tc"]) and np.array_equal(a["nt"], a0["nt"]) and
                np.array_equal(a["st"]["iters"], a0["st"]["iters"]) and np.array_equal(a["st"]["chi2"], a0["st"]["chi2"], equal_nan=True))
        bad += not same
        print(f"case {case:3d}: {W}x{Hh} L{L} N={N:4d} P={P:2d} ctx={G} chunk={chunk:3d} chained={int(chained)} padded={int(padded)} ragged={int(ragged)} "
              f"cap={prm.max_iters:2d} -> {'identical' if same else 'DIFFERENT'}", flush=True)
    print(f"{n_cases} cases, {bad} different")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
