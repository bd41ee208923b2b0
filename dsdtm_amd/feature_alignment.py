"""Feature_Alignment — host mirror of the reference class (include/Feature_alignment.h:23-99).

`Align2DGaussNewton` keeps the reference's static signature (image, bordered patch, patch,
max iterations, in/out pixel) and runs on the GPU; `align2d_batch` is the form the
speculative SearchLocalPoints needs (all candidates of a frame in one launch).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

mHalf_PatchSize = 4   # include/Feature_alignment.h:21


def align2d_batch(cur_pyr, patch_border, patch, level, px, max_iters, ctx=None):
    """M features at once. px: (M,2) float64 in level coordinates (modified copy returned).
    Returns (converged bool[M], px float64[M,2])."""
    ctx = ctx or capi.default_context()
    cur, keep = capi.pyramid_struct(cur_pyr)
    pb = np.ascontiguousarray(patch_border, np.uint8).reshape(-1, 100)
    p = np.ascontiguousarray(patch, np.uint8).reshape(-1, 64)
    lv = np.ascontiguousarray(level, np.int32).reshape(-1)
    pxa = np.array(px, dtype=np.float64).reshape(-1, 2).copy()
    m = len(lv)
    assert len(pb) == m and len(p) == m and len(pxa) == m
    conv = np.zeros(m, np.uint8)
    rc = ctx.lib.dsdtm_align2d_batch(ctx.handle, C.byref(cur), pb.ctypes.data_as(capi.u8p),
                                     p.ctypes.data_as(capi.u8p), lv.ctypes.data_as(C.POINTER(C.c_int32)),
                                     pxa.ctypes.data_as(C.POINTER(C.c_double)), conv.ctypes.data_as(capi.u8p),
                                     int(max_iters), m)
    ctx.check(rc)
    return conv.astype(bool), pxa


def warp_patches(kf_pyrs, cam, T_kf_w, T_cur_w, cand_kf, ref_px, ref_level, ref_bearing, p_world,
                 max_search_level, ctx=None):
    """SolveAffineMatrix + GetBestSearchLevel + WarpAffine + GetPatchNoBoarder for M candidates
    (src/Feature_alignment.cpp:160-275). Returns (affine[M,4], search_level[M], border[M,100], patch[M,64])."""
    ctx = ctx or capi.default_context()
    n_kf = len(kf_pyrs)
    arr = (capi.Pyramid * n_kf)()
    keep = []
    for i, p in enumerate(kf_pyrs):
        s, k = capi.pyramid_struct(p)
        arr[i] = s
        keep.append(k)
    m = len(cand_kf)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    Tk = np.ascontiguousarray(T_kf_w, np.float64).reshape(n_kf, 12)
    Tc = np.ascontiguousarray(T_cur_w, np.float64).reshape(12)
    ck = np.ascontiguousarray(cand_kf, np.int32)
    rp = np.ascontiguousarray(ref_px, np.float32).reshape(m, 2)
    rl = np.ascontiguousarray(ref_level, np.int32)
    rb = np.ascontiguousarray(ref_bearing, np.float64).reshape(m, 3)
    pw = np.ascontiguousarray(p_world, np.float64).reshape(m, 3)
    aff = np.zeros((m, 4))
    sl = np.zeros(m, np.int32)
    pb = np.zeros((m, 100), np.uint8)
    pp = np.zeros((m, 64), np.uint8)
    rc = ctx.lib.dsdtm_warp_patches(ctx.handle, arr, n_kf, C.byref(capi.camera_struct(cam)),
                                    Tk.ctypes.data_as(dp), Tc.ctypes.data_as(dp), ck.ctypes.data_as(ip),
                                    rp.ctypes.data_as(C.POINTER(C.c_float)), rl.ctypes.data_as(ip),
                                    rb.ctypes.data_as(dp), pw.ctypes.data_as(dp), int(max_search_level), m,
                                    aff.ctypes.data_as(dp), sl.ctypes.data_as(ip),
                                    pb.ctypes.data_as(capi.u8p), pp.ctypes.data_as(capi.u8p))
    ctx.check(rc)
    return aff, sl, pb, pp


def match_candidates_frames(cur_frame, keyframes, cam, T_kf_w, T_cur_w, cand_kf, ref_px, ref_level, ref_bearing, p_world,
                            cand_px, max_search_level, max_iters, ctx=None):
    """FindMatchDirect (src/Feature_alignment.cpp:142-156) for M candidates on device-resident frames in
    ONE library call: warp prelude + Align2D, the warped patches never leave the device.
    Returns (converged bool[M], px float64[M,2] in level-0 pixels, search_level int32[M])."""
    ctx = ctx or capi.default_context()
    dcur = capi.device_frame_of(ctx, cur_frame)
    dk = [capi.device_frame_of(ctx, k) for k in keyframes]
    n_kf, m = len(dk), len(cand_kf)
    handles = (C.c_void_p * n_kf)(*[d.handle for d in dk])
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    Tk = np.ascontiguousarray(T_kf_w, np.float64).reshape(n_kf, 12)
    Tc = np.ascontiguousarray(T_cur_w, np.float64).reshape(12)
    ck = np.ascontiguousarray(cand_kf, np.int32)
    rp = np.ascontiguousarray(ref_px, np.float32).reshape(m, 2)
    rl = np.ascontiguousarray(ref_level, np.int32)
    rb = np.ascontiguousarray(ref_bearing, np.float64).reshape(m, 3)
    pw = np.ascontiguousarray(p_world, np.float64).reshape(m, 3)
    px = np.array(cand_px, dtype=np.float64).reshape(m, 2).copy()
    sl = np.zeros(m, np.int32)
    conv = np.zeros(m, np.uint8)
    ctx.check(ctx.lib.dsdtm_match_candidates_frames(
        ctx.handle, dcur.handle, handles, n_kf, C.byref(capi.camera_struct(cam)), Tk.ctypes.data_as(dp), Tc.ctypes.data_as(dp),
        ck.ctypes.data_as(ip), rp.ctypes.data_as(C.POINTER(C.c_float)), rl.ctypes.data_as(ip), rb.ctypes.data_as(dp),
        pw.ctypes.data_as(dp), int(max_search_level), int(max_iters), m, px.ctypes.data_as(dp), sl.ctypes.data_as(ip),
        conv.ctypes.data_as(capi.u8p)))
    return conv.astype(bool), px, sl


class Feature_Alignment:
    """Only the parts of the class that are on the hot path are mirrored here; the
    reprojection grid (ResetGrid / ReprojectPoint / SearchLocalPoints, :22-126) lives in
    dsdtm_amd.search (host logic replaying the reference's cell order on GPU results)."""

    def __init__(self, camera, ctx=None):
        self.mCam = camera
        self._ctx = ctx

    @staticmethod
    def Align2DGaussNewton(tCurImg, tPatch_WithBoarder, tPatch, MaxIters, tCurPx, ctx=None) -> bool:
        """Reference signature (include/Feature_alignment.h:85). tCurPx (2 doubles) is updated in
        place, also on failure (:414)."""
        conv, px = align2d_batch([np.ascontiguousarray(tCurImg, np.uint8)], [tPatch_WithBoarder], [tPatch],
                                 [0], [[tCurPx[0], tCurPx[1]]], MaxIters, ctx)
        tCurPx[0], tCurPx[1] = px[0, 0], px[0, 1]
        return bool(conv[0])
