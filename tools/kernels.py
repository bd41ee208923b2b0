"""Device-resident timing of the secondary kernels (pyrDown, Align2D, warp prelude) with HIP events.
Usage: python tools/kernels.py   (MI355X)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsdtm_amd import capi, synth
from tests import helpers

dev = torch.device("cuda", 0); ctx = capi.Context(0); st = torch.cuda.Stream(device=dev)

def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

# ---- pyrDown: 2048 packed 640x480 pyramids, levels 1..3 from level 0
W, H, L, n = 640, 480, 4, 2048
ws, hs, ss, offs, nbytes = capi.pyramid_layout(W, H, L); pitch = (nbytes + 255) // 256 * 256
pyr = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, device=dev)
wa, ha, sa, oa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*ss), (C.c_size_t * L)(*offs)
t = timed(lambda: ctx.check(ctx.lib.dsdtm_pyrdown_batch_device(ctx.handle, pyr.data_ptr(), pitch, n, L, wa, ha, sa, oa, st.cuda_stream)))
rd = sum(ws[l] * hs[l] for l in range(L - 1)); wr = sum(ws[l] * hs[l] for l in range(1, L))
print(f"pyrDown: {n} images x 3 levels in {t*1e3:.3f} ms -> {n/t/1e3:.1f} k pyramids/s, {(rd+wr)*n/t/1e9:.0f} GB/s algorithmic (read {rd} B + write {wr} B per image; HBM peak 8000, achievable ~6300)")

# ---- Align2D: M features on one 1280x960 pyramid
sc_tex = np.clip(np.rint(synth.make_texture(960, 1280, 5)), 0, 255).astype(np.uint8)
pyr2 = synth.build_pyramid(sc_tex, 3)
ws2, hs2, ss2, offs2, nb2 = capi.pyramid_layout(1280, 960, 3)
packed = np.zeros(nb2, np.uint8)
for l in range(3): packed[offs2[l]:offs2[l] + ws2[l] * hs2[l]] = pyr2[l].reshape(-1)
dpyr = torch.from_numpy(packed).to(dev)
rng = np.random.default_rng(0)
base = 4096
cs = np.stack([rng.uniform(20, 1260, base), rng.uniform(20, 940, base)], 1)
pb, p = helpers.make_border_patches(pyr2[0], cs)
M = 262144
rep = M // base
pbd = torch.from_numpy(np.tile(pb, (rep, 1))).to(dev); pd = torch.from_numpy(np.tile(p, (rep, 1))).to(dev)
px0 = np.tile(cs, (rep, 1)) + rng.uniform(-1.5, 1.5, (M, 2))
pxd0 = torch.from_numpy(px0).to(dev); pxd = pxd0.clone()
lv = torch.zeros(M, dtype=torch.int32, device=dev); cv = torch.zeros(M, dtype=torch.uint8, device=dev)
img = capi.ImageDesc(); img.levels = 3
for l in range(3): img.width[l], img.height[l], img.stride[l], img.level_offset[l] = ws2[l], hs2[l], ss2[l], offs2[l]
img.bytes = nb2; img.data = dpyr.data_ptr()
def a2d():
    with torch.cuda.stream(st): pxd.copy_(pxd0, non_blocking=True)
    ctx.check(ctx.lib.dsdtm_align2d_batch_device(ctx.handle, C.byref(img), pbd.data_ptr(), pd.data_ptr(), lv.data_ptr(), pxd.data_ptr(), cv.data_ptr(), 10, M, st.cuda_stream))
t = timed(a2d)
print(f"Align2D (sums in the reference's order, the product path): {M} features (10 it cap) in {t*1e3:.3f} ms -> {M/t/1e6:.1f} M features/s, converged {cv.float().mean().item():.3f}")
if ctx.diag:                             # (DSDTM_PY_DIAG=1: the diagnostic library) DPP tree sums — flags can differ from the reference on the 0.03 px threshold
  with capi.debug_options(a2d_tree=1):
    t = timed(a2d)
  print(f"Align2D (DPP tree sums, diagnostic):                        {M} features (10 it cap) in {t*1e3:.3f} ms -> {M/t/1e6:.1f} M features/s, converged {cv.float().mean().item():.3f}")
