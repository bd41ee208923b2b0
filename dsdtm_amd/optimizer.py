"""Optimizer — host mirror of DSDTM::Optimizer::PoseOptimization (include/Optimizer.h:29,
src/Optimizer.cpp:20-101; called by Tracking::TrackWithLocalMap right after SearchLocalPoints,
src/Tracking.cpp:236) over the C ABI. The solve — every residual/Jacobian evaluation, the
trust-region iterations, the final residual norms — is one library call (dsdtm_pose_optimization,
HIP, one wavefront); what is left here is the reference's bookkeeping on MapPoint objects (:80-92).

There is no CPU path for the solve.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .frame import Config, Frame


class Optimizer:
    last_summary = None

    @staticmethod
    def PoseOptimization(tCurFrame: Frame, tIterations: int = 100, ctx: capi.Context | None = None):
        """tIterations is accepted and ignored, as in the reference (max_num_iterations = 100, :72).

        Reads tCurFrame.bearing / level / initial and tCurFrame.mvMapPoints (one entry per feature: an
        object with Get_Pose / IsBad / EraseFound, or None); writes the pose (Set_Pose, :78) and calls
        EraseFound on the map points the reference would (:80-92). Returns the solver summary."""
        ctx = ctx or capi.default_context()
        n = tCurFrame.n_features if len(tCurFrame.bearing) else 0
        mpts = list(getattr(tCurFrame, "mvMapPoints", [None] * n))
        assert len(mpts) == n, "mvMapPoints must have one entry per feature"
        use = np.zeros(n, np.uint8)
        pw = np.zeros((n, 3), np.float64)
        tvMpts = {}                                              # feature index -> MapPoint (:43, :57)
        for i in range(n):                                       # :45-65
            mp = mpts[i]
            if mp is None or mp.IsBad() or not tCurFrame.initial[i]:
                continue
            use[i] = 1
            pw[i] = mp.Get_Pose()
            tvMpts[i] = mp
        T = np.ascontiguousarray(tCurFrame.Get_Pose(), np.float64).reshape(12).copy()
        rn, sm = pose_optimization(ctx, tCurFrame.bearing, pw, tCurFrame.level, use, T)
        tCurFrame.Set_Pose(T.reshape(3, 4))                      # :78
        # :22-24: double(float threshold) / float mf
        thresh = float(np.float32(Config.Get("Optimization.LocalBAthreshhold"))) / float(np.float32(tCurFrame.mCamera.f))
        # :80-92 — the residual vector is in residual-BLOCK order while tvMpts is keyed by FEATURE index
        # (std::map::operator[] yields NULL for a missing key); the reference mixes the two and so do we
        for i in range(len(rn)):
            if rn[i] > thresh:
                mp = tvMpts.get(i)
                if mp is None:
                    continue
                if mp.IsBad():
                    continue
                mp.EraseFound()
        Optimizer.last_summary = sm
        return sm


def pose_optimization(ctx: capi.Context, bearing, p_world, level, use, T_cur_w, max_iterations: int = 100):
    """dsdtm_pose_optimization: T_cur_w (12 doubles) is updated in place; returns (residual norms in
    residual-block order, summary dict)."""
    bearing = np.ascontiguousarray(bearing, np.float64).reshape(-1, 3)
    pw = np.ascontiguousarray(p_world, np.float64).reshape(-1, 3)
    level = np.ascontiguousarray(level, np.int32)
    use = np.ascontiguousarray(use, np.uint8)
    n = len(use)
    assert len(bearing) == n and len(pw) == n and len(level) == n
    assert T_cur_w.dtype == np.float64 and T_cur_w.size == 12 and T_cur_w.flags.c_contiguous
    rn = np.zeros(max(n, 1))
    prm = capi.PoseOptParams(int(max_iterations), 0)
    sm = capi.PoseOptSummary()
    dp = C.POINTER(C.c_double)
    f = ctx.lib.dsdtm_pose_optimization
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, dp, dp, C.POINTER(C.c_int32), capi.u8p, C.c_int, dp, C.POINTER(capi.PoseOptParams), dp,
                  C.POINTER(capi.PoseOptSummary)]
    ctx.check(f(ctx.handle, bearing.ctypes.data_as(dp), pw.ctypes.data_as(dp), level.ctypes.data_as(C.POINTER(C.c_int32)),
                use.ctypes.data_as(capi.u8p), n, T_cur_w.ctypes.data_as(dp), C.byref(prm), rn.ctypes.data_as(dp),
                C.byref(sm)))
    d = sm.as_dict()
    return rn[:d["n_residual_blocks"]].copy(), d
