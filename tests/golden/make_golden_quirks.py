#!/usr/bin/env python3
"""Writes tests/golden/quirks.npz: the inputs of tests/quirk_fixtures.py (built to exercise every quirk of SURVEY.md §8.1) and
the FAITHFUL oracle's outputs for them. The reference holds no vectors for this path, so — like
make_golden.py — this pins the oracle, not the reference; what it adds is the input coverage the mutation tests need.
Re-run only when the oracle or the fixtures are deliberately changed:   python tests/golden/make_golden_quirks.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import oracle_lib, quirk_fixtures as Q  # noqa: E402


def collect():
    d = {}
    cases = Q.sparse_cases()
    base = cases["main"][0]
    for l in range(3):
        d[f"ref{l}"], d[f"cur{l}"] = base.ref_pyr[l], base.cur_pyr[l]
    d["T_ref_w"] = base.T_ref_w
    for name, case in cases.items():
        sc, prm, min_fts, T_seed = case
        d[f"{name}_px"], d[f"{name}_bearing"], d[f"{name}_p_world"], d[f"{name}_initial"] = sc.px, sc.bearing, sc.p_world, sc.initial
        d[f"{name}_params"] = np.array(list(prm) + [min_fts])
        d[f"{name}_T_seed"] = sc.T_cur_w_seed if T_seed is None else T_seed
        o = Q.sparse_outputs(lambda s, *p, **kw: oracle_lib.sparse_align(s, *p, **kw), case)
        d[f"{name}_out_T"], d[f"{name}_out_n"] = o["T"], o["n"]
        for k in ("iters", "exit_code", "n_ref", "n_vis", "chi2"):
            d[f"{name}_out_{k}"] = np.array(o[k])
    a = Q.align2d_cases()
    conv, px = oracle_lib.align2d_batch(a["pyr"], a["patch_border"], a["patch"], a["level"], a["px0"], 10)
    d.update(a2d_img=a["pyr"][0], a2d_level=a["level"], a2d_patch_border=a["patch_border"], a2d_patch=a["patch"], a2d_px0=a["px0"],
             a2d_out_conv=conv, a2d_out_px=px)
    w = Q.warp_cases()
    aff, sl, pb, pp = Q.warp_outputs(oracle_lib.warp_patches, w)
    d.update(warp_img=w["pyr"][0], warp_T_kf=w["T_kf"], warp_T_cur=np.array([g[0] for g in w["groups"]]), warp_cand_kf=w["cand_kf"],
             warp_ref_level=w["ref_level"], warp_ref_px=w["ref_px"], warp_ref_bearing=w["ref_bearing"], warp_p_world=w["p_world"],
             warp_out_affine=aff, warp_out_level=sl, warp_out_border=pb, warp_out_patch=pp)
    for name in Q.SEARCH_WORLDS:                       # the worlds are seeded generators (640x480x5x4 images): outputs only
        lst, mask = Q.search_restated(name)
        d[f"search_{name}_matches"] = np.array(lst, np.float64).reshape(-1, 5)
        d[f"search_{name}_mask_rows"] = np.packbits(mask == 255, axis=1)
    return d


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "quirks.npz"), **collect())
    print("written", os.path.join(HERE, "quirks.npz"), os.path.getsize(os.path.join(HERE, "quirks.npz")), "bytes")
