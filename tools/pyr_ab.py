"""pyrDown A/B on one box: one launch per level vs one launch per pyramid (band heights), batch and single frame.
Usage: python tools/pyr_ab.py   (MI355X)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsdtm_amd import capi

dev = torch.device("cuda", 0); ctx = capi.Context(0, diag=True)   # the diagnostic library (dsdtm_debug_* / switches); st = torch.cuda.Stream(device=dev)

def timed(fn, reps, warm=3):
    for _ in range(warm): fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

def case(W, H, L, n, reps, bands=(8, 10, 12, 15, 20, 30)):
    ws, hs, ss, offs, nbytes = capi.pyramid_layout(W, H, L); pitch = (nbytes + 255) // 256 * 256
    pyr = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, device=dev)
    wa, ha, sa, oa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*ss), (C.c_size_t * L)(*offs)
    alg = sum(ws[l] * hs[l] + ws[l + 1] * hs[l + 1] for l in range(L - 1))
    fn = lambda: ctx.check(ctx.lib.dsdtm_pyrdown_batch_device(ctx.handle, pyr.data_ptr(), pitch, n, L, wa, ha, sa, oa, st.cuda_stream))
    for name, opt in [("per level", dict(pyr_fused=0)), ("fused auto", dict(pyr_fused=2))] + [(f"fused band {b}", dict(pyr_fused=2, pyr_band=b)) for b in bands]:
        with capi.debug_options(**{"pyr_band": 0, **opt}):
            ts = [timed(fn, reps) for _ in range(3)]
        t = min(ts)
        print(f"{W}x{H} L={L} n={n:5d} {name:14s}: {t*1e3:8.4f} ms  {n/t/1e6:7.3f} M pyramids/s  {alg*n/t/1e9:7.0f} GB/s algorithmic", flush=True)

case(640, 480, 4, 2048, 20)
if os.environ.get("PYR_SMALL"):
    for n in (1, 4, 16, 64, 256, 512):
        case(640, 480, 4, n, 100, bands=())
        case(640, 480, 5, n, 100, bands=())
if os.environ.get("PYR_ALL"):
    case(640, 480, 5, 2048, 20)
    case(640, 480, 4, 1, 200)
    case(640, 480, 5, 1, 200)
    case(640, 480, 4, 16, 100)
    case(1280, 960, 4, 512, 20)
