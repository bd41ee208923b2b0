"""Detector image work (FAST-10 score map, non-max + Shi-Tomasi + best corner per cell) on batches of packed device
pyramids: device time per 640x480x5-level frame, HIP events on the launch stream.
Usage: python tools/detect_bench.py [frames ...]   (MI355X)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsdtm_amd import capi, synth

dev = torch.device("cuda", 0); ctx = capi.Context(0); st = torch.cuda.Stream(device=dev)
W, H, L = 640, 480, 5
ws, hs, ss, offs, nbytes = capi.pyramid_layout(W, H, L); pitch = (nbytes + 255) // 256 * 256
base = 8
packed = np.zeros((base, pitch), np.uint8)
for i in range(base):
    pyr = synth.build_pyramid(np.clip(np.rint(synth.make_texture(H, W, 40 + i)), 0, 255).astype(np.uint8), L)
    for l in range(L): packed[i, offs[l]:offs[l] + ws[l] * hs[l]] = pyr[l].reshape(-1)
cell, gc, gr = 25, (W + 24) // 25, (H + 24) // 25
G = gc * gr
prm = capi.DetectParams(cell, gc, gr, L, 20, 5.0)
wa, ha, sa, oa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*ss), (C.c_size_t * L)(*offs)
alg = sum(ws[l] * hs[l] for l in range(L))
for n in [int(a) for a in sys.argv[1:]] or [1, 16, 256, 1024]:
    d_pyr = torch.from_numpy(np.tile(packed, ((n + base - 1) // base, 1))[:n]).to(dev)
    d_score = torch.empty((n, pitch), dtype=torch.uint8, device=dev); d_key = torch.empty((n, G), dtype=torch.int64, device=dev)
    d_s = torch.empty((n, G), dtype=torch.float32, device=dev); d_x, d_y, d_l = (torch.empty((n, G), dtype=torch.int32, device=dev) for _ in range(3))
    fn = lambda: ctx.check(ctx.lib.dsdtm_detect_cells_batch_device(ctx.handle, d_pyr.data_ptr(), pitch, n, L, wa, ha, sa, oa, None, C.byref(prm),
                           d_score.data_ptr(), d_key.data_ptr(), d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), d_l.data_ptr(), st.cuda_stream))
    for _ in range(3): fn()
    st.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); st.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    corners = int((d_s > 5.0).sum().item())
    # bytes: the pyramid read by the score pass + the score map written, then read by the select pass
    print(f"detector, {n:5d} frames of {W}x{H}x{L} levels per call: {t*1e3:8.4f} ms = {t/n*1e6:7.3f} us per frame, "
          f"{3*alg*n/t/1e9:6.0f} GB/s algorithmic (pyramid read + score map written and read: {3*alg} B per frame), "
          f"{corners / n:.0f} cells with a corner per frame", flush=True)
