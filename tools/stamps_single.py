"""Diagnostic: ONE pair alone on the GPU (the live tracker's Run) — where its time goes, from the in-kernel stamps of the
diagnostic instantiation (never used for timing claims). Usage: python tools/stamps_single.py [cap]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsdtm_amd import capi, synth

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0); ctx = capi.Context(0, diag=True)
cam = synth.Camera.tum(640, 480); cs = capi.camera_struct(cam); prm = capi.AlignParams(4, 0, cap, 15)
st = torch.cuda.Stream(device=dev)
d = bench.build_batch(torch, dev, ctx, cam, 1, 640, 480, 4, 300, seed=0xD5D7, stream=st)
stamps = torch.zeros((52,), dtype=torch.int64, device=dev)
f = ctx.lib.dsdtm_debug_sparse_align_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(capi.BatchDesc), C.POINTER(capi.Camera), C.POINTER(capi.AlignParams), C.c_void_p, C.c_void_p]
for rep in range(5):
    d["T_cur_w"].copy_(d["T_seed"]); stamps.zero_(); torch.cuda.synchronize()
    ctx.check(f(ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stamps.data_ptr(), st.cuda_stream)); st.synchronize()
a = stamps.cpu().numpy().astype(np.float64)
s, w, pw, lv, bf, rf, rt = a[:8], a[8:20], a[20:36], a[36:44], a[44:48], a[48], a[49:52]
n_it = s[3]
wall = (rt[1] - rt[0]) / 100.0
print(f"one pair, 300 features, 4 levels, cap {cap}: {n_it:.0f} iterations, {s[4]:.0f} cycles = {wall:.1f} us ({s[4] / wall / 1e3:.2f} GHz)")
print(f"solver: first-pass waits {s[0]:.0f} ({s[0]/4:.0f} per level) | later-pass waits {s[1]:.0f} ({s[1]/max(n_it-4,1):.0f} each) | solve {s[2]:.0f} ({s[2]/n_it:.0f} each) | H refresh {rf:.0f} ({rf/4:.0f} per level)")
print(f"solve sub-phases per solve: sums+LDS {s[5]/n_it:.0f} | factor/apply {s[6]/n_it:.0f} | exp+compose+publish {s[7]/n_it:.0f}")
print(f"wave0: precompute {w[0]:.0f} ({w[0]/4:.0f} per level) | passes {w[1]:.0f} ({w[1]/n_it:.0f} each) | H-block+store {w[2]:.0f} | waits for the solver {w[3]:.0f}")
print("wave0 by level 0..3: precompute " + " ".join(f"{v:.0f}" for v in lv[:4]) + " | first pass " + " ".join(f"{v:.0f}" for v in lv[4:8]) +
      " | wait after the first pass " + " ".join(f"{v:.0f}" for v in bf))
print("pass cycles per iteration by level: " + " ".join(f"{w[4+l]/max(w[8+l],1):.0f} ({w[8+l]:.0f})" for l in range(4)))
print("per patch wave, per iteration: pass " + " ".join(f"{pw[k]/n_it:.0f}" for k in range(5)) + " | wait " + " ".join(f"{pw[8+k]/n_it:.0f}" for k in range(5)))
