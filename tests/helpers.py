"""Shared helpers for the parity tests (GPU path vs CPU oracle on identical inputs)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from dsdtm_amd import capi, synth
from dsdtm_amd.frame import Config, Frame, frames_from_scene
from dsdtm_amd.sparse_align import Sprase_ImgAlign

# north_star tolerance: pose within 1e-4 rad / 1e-4 m of the reference CPU path
TOL_RAD = 1e-4
TOL_M = 1e-4
# what we actually hold ourselves to for FP64 storage (decision parity => only rounding noise)
TIGHT_RAD = 1e-9
TIGHT_M = 1e-9


def gpu_sparse_align(scene, max_level, min_level, max_iters, min_fts=15, T_seed=None, ctx=None):
    """Runs the product path through the reference-shaped class. Returns (T 3x4, n, stats)."""
    Config.Set("Camera.Min_fts", min_fts)
    cur, ref = frames_from_scene(scene)
    if T_seed is not None:
        cur.Set_Pose(T_seed)
    al = Sprase_ImgAlign(max_level, min_level, max_iters, ctx=ctx)
    n = al.Run(cur, ref)
    return cur.Get_Pose().copy(), n, al.last_stats


def assert_pose_close(Tg, To, tol_rad=TOL_RAD, tol_m=TOL_M, what=""):
    ang, dt = synth.pose_error(Tg, To)
    assert ang <= tol_rad and dt <= tol_m, f"{what}: pose delta {ang:.3e} rad / {dt:.3e} m"
    return ang, dt


def selftest(ctx, cases: np.ndarray) -> np.ndarray:
    cases = np.ascontiguousarray(cases, np.float64).reshape(-1, 33)
    out = np.zeros((len(cases), 120))         # selftest.hip: SELFTEST_OUT
    f = ctx.lib.dsdtm_debug_selftest
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    rc = f(ctx.handle, cases.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)), len(cases))
    ctx.check(rc)
    return out


def upper21(H):
    return np.array([H[i, j] for i in range(6) for j in range(i, 6)])


def make_border_patches(img, centers, rng=None):
    """10x10 bordered + 8x8 patches cut around integer centers of `img` (u8), as
    Test/test_Feature_alignment.cpp:56-72 builds them (bilinear at subpixel centers)."""
    from scipy.ndimage import map_coordinates
    pbs, ps = [], []
    for (cx, cy) in centers:
        ys, xs = np.meshgrid(np.arange(-5, 5) + cy, np.arange(-5, 5) + cx, indexing="ij")
        pb = map_coordinates(img.astype(np.float64), [ys, xs], order=1, mode="nearest")
        pb = np.clip(np.floor(pb), 0, 255).astype(np.uint8)   # the reference truncates float->uchar
        pbs.append(pb.reshape(100))
        ps.append(pb[1:9, 1:9].reshape(64))
    return np.array(pbs), np.array(ps)


class GoldenScene:
    """AlignScene-shaped view of a tests/golden/sparse_align_*.npz fixture."""

    def __init__(self, path):
        d = np.load(path)
        self.d = d
        L = int(d["levels"])
        c = d["cam"]
        self.cam = synth.Camera(c[0], c[1], c[2], c[3], c[4], int(c[5]), int(c[6]))
        self.ref_pyr = [d[f"ref{l}"] for l in range(L)]
        self.cur_pyr = [d[f"cur{l}"] for l in range(L)]
        self.px, self.bearing, self.p_world, self.initial = d["px"], d["bearing"], d["p_world"], d["initial"]
        self.T_ref_w, self.T_cur_w_seed, self.T_cur_w_true = d["T_ref_w"], d["T_seed"], d["T_true"]
        self.params = tuple(int(x) for x in d["params"])


def golden_path(name):
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
