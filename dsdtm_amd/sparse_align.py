"""Sprase_ImgAlign — host mirror of the reference class (include/Sprase_ImageAlign.h:20-69)
over the C ABI. Same constructor arguments, same `Run(cur, ref)` argument order (current frame
first, include/Sprase_ImageAlign.h:29), same return value and side effect: the current
frame's pose is overwritten (src/Sprase_ImageAlign.cpp:57), nothing else is touched.

All arithmetic happens in libdsdtm_amd.so (HIP, gfx950). There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .frame import Config, Frame


class Sprase_ImgAlign:
    def __init__(self, tMaxLevel: int, tMinLevel: int, tMaxIterators: int, ctx: capi.Context | None = None,
                 resident_frames: bool = False):
        self.mnMaxLevel = int(tMaxLevel)
        self.mnMinLevel = int(tMinLevel)
        self.mnMaxIterators = int(tMaxIterators)
        self.mnMinfts = int(Config.Get("Camera.Min_fts"))        # src/Sprase_ImageAlign.cpp:14
        self._ctx = ctx
        self.last_stats = None
        # resident_frames: a frame's pyramid is uploaded once (dsdtm_frame, cached on the Frame object)
        # and reused by every Run it takes part in — in tracking each frame is `cur` once and `ref` once
        self.resident_frames = bool(resident_frames)

    def _context(self):
        if self._ctx is None:
            self._ctx = capi.default_context()
        return self._ctx

    @staticmethod
    def _device_frame(ctx, frame: Frame):
        return capi.device_frame_of(ctx, frame)

    def Run(self, tCurFrame: Frame, tRefFrame: Frame) -> int:
        ctx = self._context()
        if self.resident_frames:
            return self._run_frames(ctx, tCurFrame, tRefFrame)
        ref, k1 = capi.pyramid_struct(tRefFrame.mvImg_Pyr)
        cur, k2 = capi.pyramid_struct(tCurFrame.mvImg_Pyr)
        cam = capi.camera_struct(tRefFrame.mCamera)
        n = tRefFrame.n_features
        dp = C.POINTER(C.c_double)
        Tr = np.ascontiguousarray(tRefFrame.Get_Pose(), np.float64).reshape(12).copy()
        Tc = np.ascontiguousarray(tCurFrame.Get_Pose(), np.float64).reshape(12).copy()
        prm = capi.AlignParams(self.mnMaxLevel, self.mnMinLevel, self.mnMaxIterators, self.mnMinfts)
        nt = C.c_int(0)
        st = capi.AlignStats()
        rc = ctx.lib.dsdtm_sparse_align(
            ctx.handle, C.byref(ref), C.byref(cur), C.byref(cam),
            tRefFrame.px.ctypes.data_as(C.POINTER(C.c_float)), tRefFrame.bearing.ctypes.data_as(dp),
            tRefFrame.p_world.ctypes.data_as(dp), tRefFrame.initial.ctypes.data_as(capi.u8p), n,
            Tr.ctypes.data_as(dp), Tc.ctypes.data_as(dp), C.byref(prm), C.byref(nt), C.byref(st))
        ctx.check(rc)
        self.last_stats = st.as_dict()
        if n >= self.mnMinfts and self.mnMaxLevel - 1 >= self.mnMinLevel:
            tCurFrame.Set_Pose(Tc.reshape(3, 4))                  # :57
        return nt.value                                           # :59

    def _run_frames(self, ctx, tCurFrame: Frame, tRefFrame: Frame) -> int:
        dref, dcur = self._device_frame(ctx, tRefFrame), self._device_frame(ctx, tCurFrame)
        cam = capi.camera_struct(tRefFrame.mCamera)
        n = tRefFrame.n_features
        dp = C.POINTER(C.c_double)
        Tr = np.ascontiguousarray(tRefFrame.Get_Pose(), np.float64).reshape(12).copy()
        Tc = np.ascontiguousarray(tCurFrame.Get_Pose(), np.float64).reshape(12).copy()
        prm = capi.AlignParams(self.mnMaxLevel, self.mnMinLevel, self.mnMaxIterators, self.mnMinfts)
        nt = C.c_int(0)
        st = capi.AlignStats()
        ctx.check(ctx.lib.dsdtm_sparse_align_frames(
            ctx.handle, dref.handle, dcur.handle, C.byref(cam),
            tRefFrame.px.ctypes.data_as(C.POINTER(C.c_float)), tRefFrame.bearing.ctypes.data_as(dp),
            tRefFrame.p_world.ctypes.data_as(dp), tRefFrame.initial.ctypes.data_as(capi.u8p), n,
            Tr.ctypes.data_as(dp), Tc.ctypes.data_as(dp), C.byref(prm), C.byref(nt), C.byref(st)))
        self.last_stats = st.as_dict()
        if n >= self.mnMinfts and self.mnMaxLevel - 1 >= self.mnMinLevel:
            tCurFrame.Set_Pose(Tc.reshape(3, 4))                  # :57
        return nt.value                                           # :59
