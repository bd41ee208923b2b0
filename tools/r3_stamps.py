"""Diagnostic: cycle stamps of the three-slot kernel (separate instantiation; never used for timing claims).
Usage: python tools/r3_stamps.py [--pairs N]"""
import argparse, ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsdtm_amd import capi, synth

ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=3072); a = ap.parse_args()
dev = torch.device("cuda", 0); ctx = capi.Context(0)
cam = synth.Camera.tum(640, 480); cs = capi.camera_struct(cam); prm = capi.AlignParams(4, 0, 10, 15)
st = torch.cuda.Stream(device=dev)
d = bench.build_batch(torch, dev, ctx, cam, a.pairs, 640, 480, 4, 300, seed=0xD5D7, stream=st)
stamps = torch.zeros((a.pairs * 16,), dtype=torch.int64, device=dev)
f = ctx.lib.dsdtm_debug_sparse_align_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(capi.BatchDesc), C.POINTER(capi.Camera), C.POINTER(capi.AlignParams), C.c_void_p, C.c_void_p]
with capi.debug_options(reg_slots=3):
    for rep in range(3):
        d["T_cur_w"].copy_(d["T_seed"]); stamps.zero_(); torch.cuda.synchronize()
        ctx.check(f(ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stamps.data_ptr(), st.cuda_stream)); st.synchronize()
s = stamps.cpu().numpy().astype(np.float64).reshape(a.pairs, 16)
n = s[:, 6]
print("three-slot kernel, %d pairs, s_memtime stamps (same unit as tools/stamps.py)" % a.pairs)
print("iterations per pair %.2f | pair total %.0f" % (n.mean(), s[:, 0].mean()))
print("lead wave, per pair: passes %.0f | waits for the slot's arrivals %.0f | waits for the helper %.0f | solves %.0f | level starts %.0f" %
      tuple(s[:, i].mean() for i in (1, 2, 3, 4, 5)))
print("lead wave, per iteration: pass %.0f | arrivals %.0f | helper %.0f | solve %.0f ; per level start %.0f" %
      ((s[:, 1] / n).mean(), (s[:, 2] / n).mean(), (s[:, 3] / n).mean(), (s[:, 4] / n).mean(), s[:, 5].mean() / 4))
print("wave 0, per iteration: pass %.0f | wait for the lead %.0f ; per level start %.0f" %
      ((s[:, 8] / n).mean(), (s[:, 9] / n).mean(), s[:, 10].mean() / 4))
