"""Feature_detector — host mirror of the reference class (include/Feature_detection.h:35-75,
src/Feature_detection.cpp) over the C ABI. The image work of detect() — FAST-10 corners, scores,
3x3 non-maximum suppression on every pyramid level, Shi-Tomasi score, best corner per grid cell
(:75-108) — is one library call (dsdtm_detect_cells[_frame], HIP); what is left here is the
reference's order-dependent bookkeeping over at most grid_cols*grid_rows corners (:110-153): sort by
score, skip masked pixels, add the feature, paint the mask disc, stop at Camera.Max_fts.

There is no CPU path for the image work.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import capi
from .frame import Config, Frame
from .search import fill_circle


class Feature_detector:
    def __init__(self, width: int, height: int, ctx: capi.Context | None = None):
        self.mCell_size = int(Config.Get("Camera.CellSize"))             # :12
        self.mPyr_levels = int(Config.Get("Camera.MaxPyraLevels"))       # :13
        self.mMax_fts = int(Config.Get("Camera.Max_fts"))                # :14
        self.mImg_width, self.mImg_height = int(width), int(height)      # :15-16 (Camera.width / Camera.height)
        self.mGrid_rows = int(math.ceil(1.0 * self.mImg_height / self.mCell_size))   # :18
        self.mGrid_cols = int(math.ceil(1.0 * self.mImg_width / self.mCell_size))    # :19
        self.mvGrid_occupy = np.zeros(self.mGrid_rows * self.mGrid_cols, np.uint8)   # :20
        self._ctx = ctx
        self.last_cells = None

    def _context(self):
        if self._ctx is None:
            self._ctx = capi.default_context()
        return self._ctx

    # :32-38, :40-59
    def Get_CellIndex(self, x: int, y: int, level: int) -> int:
        scale = 1 << level
        return (scale * y) // self.mCell_size * self.mGrid_cols + (scale * x) // self.mCell_size

    def Set_CellIndexOccupy(self, px):
        self.mvGrid_occupy[int(px[1] / self.mCell_size) * self.mGrid_cols + int(px[0] / self.mCell_size)] = 1

    def Set_ExistingFeatures(self, features_px):
        self.mvGrid_occupy[:] = 0
        for px in np.asarray(features_px, np.float32).reshape(-1, 2):
            self.Set_CellIndexOccupy(px)

    def ResetGrid(self):                                                  # :64-67
        self.mvGrid_occupy[:] = 0

    def detect_cells(self, frame: Frame, detection_threshold: float):
        """The per-cell corners of :74-108 as (score f32, x, y, level) arrays of length rows*cols."""
        ctx = self._context()
        G = self.mGrid_rows * self.mGrid_cols
        levels = min(self.mPyr_levels, len(frame.mvImg_Pyr))
        prm = capi.DetectParams(self.mCell_size, self.mGrid_cols, self.mGrid_rows, levels, 20, float(detection_threshold))
        score = np.zeros(G, np.float32)
        cx, cy, cl = np.zeros(G, np.int32), np.zeros(G, np.int32), np.zeros(G, np.int32)
        ip = C.POINTER(C.c_int32)
        args = (self.mvGrid_occupy.ctypes.data_as(capi.u8p), C.byref(prm), score.ctypes.data_as(C.POINTER(C.c_float)),
                cx.ctypes.data_as(ip), cy.ctypes.data_as(ip), cl.ctypes.data_as(ip))
        df = getattr(frame, "_device_frame", None)
        if df is not None and df.ctx is ctx and df.handle is not None:
            ctx.check(ctx.lib.dsdtm_detect_cells_frame(ctx.handle, df.handle, *args))
        else:
            pyr, keep = capi.pyramid_struct(frame.mvImg_Pyr)
            ctx.check(ctx.lib.dsdtm_detect_cells(ctx.handle, C.byref(pyr), *args))
        self.last_cells = (score, cx, cy, cl)
        return self.last_cells

    def detect(self, frame: Frame, detection_threshold: float, tFirst: bool = True):
        """void Feature_detector::detect(Frame*, const double, const bool) — :69-154. New features are
        appended to the frame (pixel + level; bearing left to the caller as with Add_Feature(.., 0), :144).
        `frame.mDynamicMask` (optional, u8, level-0 size) is Frame::mDynamicMask, the moving-object mask
        Set_Mask subtracts (src/Frame.cpp:294-296). Corners of EQUAL score keep their cell order here (stable
        sort); the reference's std::sort (:110) leaves the order of ties unspecified."""
        if frame.n_features >= self.mMax_fts:                             # :71-72
            return 0
        score, cx, cy, cl = self.detect_cells(frame, detection_threshold)
        order = np.argsort(-score, kind="stable")                          # :110 std::sort by descending score
        mask = np.full((frame.mvImg_Pyr[0].shape[0], frame.mvImg_Pyr[0].shape[1]), 255, np.uint8)   # src/Frame.cpp:64
        if frame.n_features > 0:                                          # :119-122 Frame::Set_Mask (src/Frame.cpp:286-298)
            rad = int(Config.Get("Camera.Min_dist"))
            for k in range(frame.n_features):
                if frame.initial[k]:                                      # features that have a map point
                    fill_circle(mask, int(round(float(frame.px[k, 0]))), int(round(float(frame.px[k, 1]))), rad, 0)
            # Set_Mask also removes the moving-object mask (src/Frame.cpp:294-296): mDynamicMask thresholded at 200,
            # mImgMask - mDynamicMask with cv::Mat's saturating u8 subtraction => 0 wherever the object mask is set
            dyn = getattr(frame, "mDynamicMask", None)
            if dyn is not None:
                mask[np.asarray(dyn, np.uint8) > 200] = 0
        new_px, new_level = [], []
        n = frame.n_features
        for k in order:                                                   # :124-150
            if score[k] > 20:
                x, y = int(cx[k]), int(cy[k])
                if mask[y, x] != 255:                                     # :139
                    continue
                new_px.append((x, y)); new_level.append(int(cl[k]))       # :142
                fill_circle(mask, x, y, self.mCell_size, 0)               # :143
                n += 1
            if n >= self.mMax_fts:                                        # :148-149
                break
        self.ResetGrid()                                                  # :152
        if new_px:
            m = len(new_px)
            frame.set_features(np.concatenate([frame.px, np.asarray(new_px, np.float32)]),
                               np.concatenate([frame.bearing, np.zeros((m, 3))]),
                               np.concatenate([frame.p_world, np.zeros((m, 3))]),
                               np.concatenate([frame.initial, np.zeros(m, np.uint8)]),
                               np.concatenate([frame.level, np.asarray(new_level, np.int32)]))
        return len(new_px)
