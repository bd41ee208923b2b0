"""Diagnostic: where a Gauss-Newton iteration spends its cycles (in-kernel stamps, separate
kernel instantiation; never used for timing claims). Usage: python tools/stamps.py [--pairs N]"""
import argparse, ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsdtm_amd import capi, synth

ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=1024); a = ap.parse_args()
dev = torch.device("cuda", 0); ctx = capi.Context(0)
cam = synth.Camera.tum(640, 480); cs = capi.camera_struct(cam); prm = capi.AlignParams(4, 0, 10, 15)
st = torch.cuda.Stream(device=dev)
d = bench.build_batch(torch, dev, ctx, cam, a.pairs, 640, 480, 4, 300, seed=0xD5D7, stream=st)
stamps = torch.zeros((a.pairs * 12,), dtype=torch.int64, device=dev)
f = ctx.lib.dsdtm_debug_sparse_align_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(capi.BatchDesc), C.POINTER(capi.Camera), C.POINTER(capi.AlignParams), C.c_void_p, C.c_void_p]
for rep in range(3):
    d["T_cur_w"].copy_(d["T_seed"]); torch.cuda.synchronize()
    ctx.check(f(ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stamps.data_ptr(), st.cuda_stream)); st.synchronize()
allst = stamps.cpu().numpy().astype(np.float64)
s = allst[:a.pairs*8].reshape(a.pairs, 8)
w = allst[a.pairs*8:].reshape(a.pairs, 4)
n_it = s[:, 3]
print("pairs", a.pairs, "iterations/pair mean", n_it.mean())
print("cycles per block: total %.0f | first-pass waits (4 levels) %.0f | later-pass waits %.0f | solve %.0f" % (s[:,4].mean(), s[:,0].mean(), s[:,1].mean(), s[:,2].mean()))
print("per first pass %.0f | per later pass %.0f | per solve %.0f cycles" % ((s[:,0]/4).mean(), (s[:,1]/np.maximum(n_it-4,1)).mean(), (s[:,2]/n_it).mean()))
print("solve sub-phases per solve: sums+LDS %.0f | factor/apply %.0f | exp+compose+publish %.0f" % ((s[:,5]/n_it).mean(), (s[:,6]/n_it).mean(), (s[:,7]/n_it).mean()))
print("wave0: precompute %.0f (per level %.0f) | passes %.0f (per pass %.0f) | H-block+store %.0f | barriers(wait for solver) %.0f" % (w[:,0].mean(), (w[:,0]/4).mean(), w[:,1].mean(), (w[:,1]/n_it).mean(), w[:,2].mean(), w[:,3].mean()))
