"""Soak of FindMatchDirect's kernels against the CPU oracle over candidate COUNTS (group boundaries of the warp prelude: 2 and 16
per group, the 8192 switch between them; 16 features per Align2D group) and random invalid candidates: affine, search level,
patch bytes, convergence flags and refined pixels bit for bit.
    python tools/soak_fmd.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dsdtm_amd import capi, synth  # noqa: E402
from tests import oracle_lib  # noqa: E402


SIZES = (1, 2, 3, 15, 16, 17, 31, 33, 127, 1000, 4097, 8191, 8192, 8193, 9001)


def main(sizes=SIZES):
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    rng = np.random.default_rng(7)
    W, Hh, L, n_fr = 320, 240, 4, 3
    ws, hs, ss, offs, nb = capi.pyramid_layout(W, Hh, L)
    pitch = (nb + 255) // 256 * 256
    scenes = [synth.make_scene(width=W, height=Hh, levels=L, n_patches=600, seed=70 + i, margin=16) for i in range(n_fr)]
    cam = scenes[0].cam
    cur_pack, kf_pack = np.zeros((n_fr, pitch), np.uint8), np.zeros((n_fr, pitch), np.uint8)
    for i, sc in enumerate(scenes):
        for l in range(L):
            kf_pack[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.ref_pyr[l].reshape(-1)
            cur_pack[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.cur_pyr[l].reshape(-1)
    Tk = np.stack([sc.T_ref_w.reshape(12) for sc in scenes])
    Tc = np.stack([sc.T_cur_w_true.reshape(12) for sc in scenes])
    tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_cur, d_kf, d_Tk, d_Tc = tdev(cur_pack), tdev(kf_pack), tdev(Tk), tdev(Tc)
    wa, ha, sa, oa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*ss), (C.c_size_t * L)(*offs)
    cs = capi.camera_struct(cam)
    bad = 0
    for m in sizes:
        fr = rng.integers(0, n_fr, m).astype(np.int32)
        idx = rng.integers(0, 600, m)
        rp = np.stack([scenes[f].px[i] for f, i in zip(fr, idx)]).astype(np.float32)
        rb = np.stack([scenes[f].bearing[i] for f, i in zip(fr, idx)])
        pw = np.stack([scenes[f].p_world[i] for f, i in zip(fr, idx)])
        rl = rng.integers(0, 2, m).astype(np.int32)
        kf = fr.copy()
        # a few invalid candidates: keyframe / level / frame out of range
        inv = rng.random(m) < 0.03
        kf_in = kf.copy(); rl_in = rl.copy(); fr_in = fr.copy()
        which = rng.integers(0, 3, m)
        kf_in[inv & (which == 0)] = 99; rl_in[inv & (which == 1)] = -1; fr_in[inv & (which == 2)] = n_fr + 5
        Xc = np.einsum("mij,mj->mi", Tc[fr].reshape(m, 3, 4)[:, :, :3], pw) + Tc[fr].reshape(m, 3, 4)[:, :, 3]
        cpx = np.stack([cam.fx * Xc[:, 0] / Xc[:, 2] + cam.cx, cam.fy * Xc[:, 1] / Xc[:, 2] + cam.cy], 1) + rng.uniform(-1, 1, (m, 2))
        d_px = tdev(cpx.copy())
        d_sl, d_cv = torch.full((m,), 7, dtype=torch.int32, device=dev), torch.ones(m, dtype=torch.uint8, device=dev)
        d_scr = torch.zeros(ctx.lib.dsdtm_match_candidates_scratch_bytes(m), dtype=torch.uint8, device=dev)
        k_fr, k_kf, k_rp, k_rl, k_rb, k_pw = tdev(fr_in), tdev(kf_in), tdev(rp), tdev(rl_in), tdev(rb), tdev(pw)   # alive across the call
        ctx.check(ctx.lib.dsdtm_match_candidates_batch_device(
            ctx.handle, d_cur.data_ptr(), n_fr, d_kf.data_ptr(), n_fr, pitch, L, wa, ha, sa, oa, C.byref(cs), d_Tk.data_ptr(), d_Tc.data_ptr(),
            k_fr.data_ptr(), k_kf.data_ptr(), k_rp.data_ptr(), k_rl.data_ptr(), k_rb.data_ptr(), k_pw.data_ptr(),
            L - 3, 10, m, d_scr.data_ptr(), d_px.data_ptr(), d_sl.data_ptr(), d_cv.data_ptr(), None))
        torch.cuda.synchronize()
        g_px, g_sl, g_cv = d_px.cpu().numpy(), d_sl.cpu().numpy(), d_cv.cpu().numpy().astype(bool)
        # oracle, per current frame (its entry takes one current pose), valid candidates only
        ok = ~inv
        o_px, o_sl, o_cv = cpx.copy(), np.full(m, -1, np.int32), np.zeros(m, bool)
        for f in range(n_fr):
            sel = np.nonzero(ok & (fr == f))[0]
            if not len(sel):
                continue
            aff, sl, pb, pp = oracle_lib.warp_patches([sc.ref_pyr for sc in scenes], cam, Tk.reshape(n_fr, 3, 4), Tc[f].reshape(3, 4), kf[sel], rp[sel],
                                                      rl[sel], rb[sel], pw[sel], L - 3)
            cv, px = oracle_lib.align2d_batch(scenes[f].cur_pyr, pb, pp, sl, cpx[sel] / (1 << sl)[:, None], 10)
            o_px[sel] = px * (1 << sl)[:, None]; o_sl[sel] = sl; o_cv[sel] = cv
        same = np.array_equal(g_sl, o_sl) and np.array_equal(g_cv, o_cv) and np.array_equal(g_px, o_px, equal_nan=True)
        bad += not same
        if not same:
            dsl, dcv = np.nonzero(g_sl != o_sl)[0], np.nonzero(g_cv != o_cv)[0]
            dpx = np.nonzero(~((g_px == o_px) | (np.isnan(g_px) & np.isnan(o_px))).all(1))[0]
            print("   level differs at", dsl[:5], g_sl[dsl[:5]], o_sl[dsl[:5]], "| flag at", dcv[:5], "| pixel at", dpx[:5], g_px[dpx[:3]].tolist(), o_px[dpx[:3]].tolist(),
                  "inv there:", inv[dpx[:5]])
        print(f"m = {m:5d}: {int(inv.sum()):3d} invalid, {int(o_cv.sum()):5d} matched -> {'identical' if same else 'DIFFERENT'}", flush=True)
    print("differences:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
