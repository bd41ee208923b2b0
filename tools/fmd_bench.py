"""FindMatchDirect for many current frames (warp prelude + Align2D): device time per call by workgroup shape of the warp
prelude (DSDTM_WARP_GROUP candidates per 128-thread group) — bench.py's FindMatchDirect entry without the rest of the line.
    python tools/fmd_bench.py [groups ...]          default: 0 (auto) 2 8 16 32 64
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dsdtm_amd import capi, synth  # noqa: E402


def main():
    groups = [int(v) for v in sys.argv[1:]] or [0, 2, 8, 16, 32, 64]
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, diag=True)   # the diagnostic library (dsdtm_debug_* / switches)
    stream = torch.cuda.Stream(device=dev)
    tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for nfm, ncand in ((64, 800), (1, 816)):
        Wm, Hm, Lm = 640, 480, 5
        ws, hs, ss, offs, nb = capi.pyramid_layout(Wm, Hm, Lm)
        pitch = (nb + 255) // 256 * 256
        nsc = min(8, nfm)
        cur_pack, kf_pack, cols = np.zeros((nsc, pitch), np.uint8), np.zeros((nsc, pitch), np.uint8), []
        Tk8, Tc8 = np.zeros((nsc, 12)), np.zeros((nsc, 12))
        for i in range(nsc):
            scn = synth.make_scene(width=Wm, height=Hm, levels=Lm, n_patches=ncand, seed=900 + i, margin=30)
            for l in range(Lm):
                kf_pack[i, offs[l]:offs[l] + ws[l] * hs[l]] = scn.ref_pyr[l].reshape(-1)
                cur_pack[i, offs[l]:offs[l] + ws[l] * hs[l]] = scn.cur_pyr[l].reshape(-1)
            Tk8[i], Tc8[i] = scn.T_ref_w.reshape(12), scn.T_cur_w_true.reshape(12)
            Xc = scn.p_world @ scn.T_cur_w_true[:, :3].T + scn.T_cur_w_true[:, 3]
            pxc = np.stack([scn.cam.fx * Xc[:, 0] / Xc[:, 2] + scn.cam.cx, scn.cam.fy * Xc[:, 1] / Xc[:, 2] + scn.cam.cy], 1)
            cols.append((scn.px, scn.bearing, scn.p_world, pxc + np.random.default_rng(i).uniform(-1.0, 1.0, pxc.shape)))
        cam_m = capi.camera_struct(scn.cam)
        rep_ = nfm // nsc
        Mm = nfm * ncand
        d_cur, d_kf = tdev(np.tile(cur_pack, (rep_, 1))), tdev(np.tile(kf_pack, (rep_, 1)))
        d_Tk, d_Tc = tdev(np.tile(Tk8, (rep_, 1))), tdev(np.tile(Tc8, (rep_, 1)))
        fr_idx = np.repeat(np.arange(nfm, dtype=np.int32), ncand)
        d_fr, d_kfi = tdev(fr_idx), tdev(fr_idx.copy())
        d_rp = tdev(np.tile(np.concatenate([c[0] for c in cols]), (rep_, 1)).astype(np.float32))
        d_rl = torch.zeros(Mm, dtype=torch.int32, device=dev)
        d_rb, d_pw = tdev(np.tile(np.concatenate([c[1] for c in cols]), (rep_, 1))), tdev(np.tile(np.concatenate([c[2] for c in cols]), (rep_, 1)))
        d_px0 = tdev(np.tile(np.concatenate([c[3] for c in cols]), (rep_, 1)))
        d_px = d_px0.clone()
        d_sl, d_cv = torch.zeros(Mm, dtype=torch.int32, device=dev), torch.zeros(Mm, dtype=torch.uint8, device=dev)
        d_scr = torch.empty(ctx.lib.dsdtm_match_candidates_scratch_bytes(Mm), dtype=torch.uint8, device=dev)
        wa, ha, sa, oa = (C.c_int * Lm)(*ws), (C.c_int * Lm)(*hs), (C.c_int * Lm)(*ss), (C.c_size_t * Lm)(*offs)

        def fmd():
            ctx.check(ctx.lib.dsdtm_match_candidates_batch_device(
                ctx.handle, d_cur.data_ptr(), nfm, d_kf.data_ptr(), nfm, pitch, Lm, wa, ha, sa, oa, C.byref(cam_m), d_Tk.data_ptr(), d_Tc.data_ptr(),
                d_fr.data_ptr(), d_kfi.data_ptr(), d_rp.data_ptr(), d_rl.data_ptr(), d_rb.data_ptr(), d_pw.data_ptr(), Lm - 3, 10, Mm,
                d_scr.data_ptr(), d_px.data_ptr(), d_sl.data_ptr(), d_cv.data_ptr(), stream.cuda_stream))
        ref = None
        for g in groups:
            ctx.check(ctx.lib.dsdtm_debug_set_option(b"warp_group", g))
            ev = []
            with torch.cuda.stream(stream):
                for k in range(33):
                    d_px.copy_(d_px0, non_blocking=True)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream); fmd(); b.record(stream)
                    ev.append((a, b))
            stream.synchronize()
            t = [a.elapsed_time(b) for a, b in ev][3:]
            out = (d_px.cpu().numpy().copy(), d_sl.cpu().numpy().copy(), d_cv.cpu().numpy().copy())
            same = ref is None or all(np.array_equal(x, y, equal_nan=True) for x, y in zip(out, ref))
            ref = ref or out
            print(f"{nfm} frames x {ncand} candidates, warp group {g:2d}: {np.median(t) * 1e3:7.1f} us per call (min {np.min(t) * 1e3:.1f}); "
                  f"{Mm / np.median(t) / 1e3:.0f} M candidates/s; outputs identical to the first row: {same}", flush=True)
        ctx.check(ctx.lib.dsdtm_debug_set_option(b"warp_group", 0))


if __name__ == "__main__":
    main()
