#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for them.

The reference (gaochq/DSDTM) holds no golden vectors for this path and cannot be built here
(SURVEY.md §8c), so these fixtures pin the ORACLE (oracle/dsdtm_oracle.c): the CPU suite checks
that the oracle still reproduces them bit for bit, the GPU suite checks the HIP path against the
same numbers. Re-run only when the oracle is deliberately changed:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dsdtm_amd import synth  # noqa: E402
from tests import helpers, oracle_lib  # noqa: E402


def sparse_case(seed, width, height, levels, n, xi, params):
    sc = synth.make_scene(width=width, height=height, levels=levels, n_patches=n, seed=seed, xi=xi, margin=10,
                          T_ref_w=synth.random_pose(np.random.default_rng(seed)), frac_uninitial=0.08)
    T, nt, st = oracle_lib.sparse_align(sc, *params)
    d = {f"ref{l}": sc.ref_pyr[l] for l in range(levels)}
    d.update({f"cur{l}": sc.cur_pyr[l] for l in range(levels)})
    d.update(px=sc.px, bearing=sc.bearing, p_world=sc.p_world, initial=sc.initial, T_ref_w=sc.T_ref_w,
             T_seed=sc.T_cur_w_seed, T_true=sc.T_cur_w_true, cam=np.array([sc.cam.fx, sc.cam.fy, sc.cam.cx, sc.cam.cy, sc.cam.f, width, height]),
             params=np.array(params), levels=levels, out_T=T, out_n=nt, out_iters=np.array(st["iters"]),
             out_exit=np.array(st["exit_code"]), out_chi2=np.array(st["chi2"]), out_nref=np.array(st["n_ref"]),
             out_nvis=np.array(st["n_vis"]))
    return d


def main():
    np.savez_compressed(os.path.join(HERE, "sparse_align_a.npz"),
                        **sparse_case(11, 160, 120, 3, 70, (0.006, -0.004, 0.003, 0.003, -0.002, 0.004), (3, 0, 10, 15)))
    np.savez_compressed(os.path.join(HERE, "sparse_align_b.npz"),
                        **sparse_case(12, 200, 152, 4, 90, (-0.008, 0.005, -0.006, -0.004, 0.003, 0.002), (4, 1, 30, 15)))
    # Align2D + pyrDown + warp prelude on one small texture
    rng = np.random.default_rng(21)
    tex = np.clip(np.rint(synth.make_texture(96, 128, 31)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 3)
    m = 24
    level = rng.integers(0, 2, m).astype(np.int32)
    pbs, ps, px0 = [], [], []
    for i in range(m):
        img = pyr[level[i]]
        c = (rng.uniform(10, img.shape[1] - 10), rng.uniform(10, img.shape[0] - 10))
        pb, p = helpers.make_border_patches(img, [c])
        pbs.append(pb[0]); ps.append(p[0]); px0.append([c[0] + rng.uniform(-1.5, 1.5), c[1] + rng.uniform(-1.5, 1.5)])
    conv, pxo = oracle_lib.align2d_batch(pyr, pbs, ps, level, np.array(px0), 10)
    cam = synth.Camera.tum(128, 96)
    T_kf = np.array([synth.random_pose(rng, 0.2, 0.05) for _ in range(2)])
    T_cur = synth.random_pose(rng, 0.2, 0.05)
    ck = rng.integers(0, 2, m).astype(np.int32)
    rl = rng.integers(0, 2, m).astype(np.int32)
    rp = np.stack([rng.uniform(16, 112, m), rng.uniform(16, 80, m)], 1).astype(np.float32)
    rb = synth.bearing_from_px(cam, rp)
    depth = rng.uniform(0.5, 3.0, m)
    pw = np.array([T_kf[ck[i]][:, :3].T @ (rb[i] * depth[i] - T_kf[ck[i]][:, 3]) for i in range(m)])
    aff, sl, wb, wp = oracle_lib.warp_patches([pyr, pyr], cam, T_kf, T_cur, ck, rp, rl, rb, pw, 2)
    np.savez_compressed(os.path.join(HERE, "align2d_pyr_warp.npz"), tex=tex, pyr1=pyr[1], pyr2=pyr[2], level=level,
                        patch_border=np.array(pbs), patch=np.array(ps), px0=np.array(px0), out_conv=conv, out_px=pxo,
                        cam=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.f, 128, 96]), T_kf=T_kf, T_cur=T_cur,
                        cand_kf=ck, ref_level=rl, ref_px=rp, ref_bearing=rb, p_world=pw, out_affine=aff,
                        out_search_level=sl, out_warp_border=wb, out_warp_patch=wp)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
