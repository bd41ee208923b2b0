"""One tracked frame of DSDTM::Tracking (reference src/Tracking.cpp:199-256) through ONE library call — the host mirror
of dsdtm_track_frame, with the side effects the reference's classes leave on the frame and the map:

    TrackWithLastFrame   cur.Set_Pose(last.Get_Pose()); Sprase_ImgAlign::Run(cur, last)                  (:199-217)
    UpdateLocalMap       ResetGrid; ReprojectPoint for every local map point                              (:258-312)
    TrackWithLocalMap    SearchLocalPoints(cur) (new Features, IncreaseFound, mask discs);
                         Optimizer::PoseOptimization(cur) (Set_Pose, the EraseFound walk)                 (:219-256)

The four-call chain (sparse_align.Sprase_ImgAlign, search.LocalPointSearch, optimizer.Optimizer on device-resident frames)
does the same with a host round trip between the steps; tests hold the two to the same bits. No CPU path: the library call
fails loudly without the HIP library or a gfx950 device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi, search
from .frame import Config, Frame


def flatten_local_map(keyframes, map_points):
    """The local map as dsdtm_track_desc wants it: map-point columns + observations in CSR form (iteration order of
    mObservations = keyframe-index order, as search.get_closest_obs walks it), each observation carrying the observing
    feature's mpx / mlevel / mNormal."""
    M = len(map_points)
    pw = np.zeros((M, 3), np.float64)
    found = np.zeros(M, np.int32)
    bad = np.zeros(M, np.uint8)
    off = np.zeros(M + 1, np.int32)
    okf, ofe = [], []
    for i, mp in enumerate(map_points):
        pw[i] = mp.Get_Pose()
        found[i] = mp.Get_FoundNums()
        bad[i] = 1 if mp.IsBad() else 0
        for k in sorted(mp.mObservations):
            okf.append(k)
            ofe.append(mp.mObservations[k])
        off[i + 1] = len(okf)
    okf = np.array(okf, np.int32)
    n = len(okf)
    opx, olv, ob = np.zeros((n, 2), np.float32), np.zeros(n, np.int32), np.zeros((n, 3), np.float64)
    for j in range(n):
        kf = keyframes[int(okf[j])]
        opx[j], olv[j], ob[j] = kf.px[ofe[j]], kf.level[ofe[j]], kf.bearing[ofe[j]]
    return dict(pw=pw, found=found, bad=bad, off=off, okf=okf, opx=opx, olv=olv, ob=ob)


class TrackCall:
    """One prepared dsdtm_track_frame call: the descriptor and every array it points at are built once (`__init__`), `run()`
    is the library call alone — what a C++ Tracking pays per frame (bench_tracking.py times it; a frame's release is the
    caller's: result["frame"].close())."""

    def __init__(self, ctx: capi.Context, cam, image, levels, last: Frame, T_seed, align, min_tracked, keyframes, map_points,
                 mask=None, cell_size=None, max_pyr_levels=None, max_matches=200, align2d_iters=10, po_iterations=100, flat=None):
        self.ctx = ctx
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 2 or image.strides[1] != 1 or image.strides[0] < image.shape[1]:
            image = np.ascontiguousarray(image, np.uint8)      # (a row-strided uint8 view goes down as it is: d.stride)
        cell_size = int(Config.Get("Camera.CellSize") if cell_size is None else cell_size)
        max_pyr_levels = int(Config.Get("Camera.MaxPyraLevels") if max_pyr_levels is None else max_pyr_levels)
        fm = flat if flat is not None else flatten_local_map(keyframes, map_points)
        d = capi.TrackDesc()
        d.image, d.width, d.height, d.stride, d.levels = image.ctypes.data, image.shape[1], image.shape[0], image.strides[0], int(levels)
        dref = capi.device_frame_of(ctx, last)
        d.ref = dref.handle
        px = np.ascontiguousarray(last.px, np.float32)
        bear, pw, ini = np.ascontiguousarray(last.bearing), np.ascontiguousarray(last.p_world), np.ascontiguousarray(last.initial, np.uint8)
        d.ref_px_xy, d.ref_bearing, d.ref_p_world, d.ref_initial = px.ctypes.data, bear.ctypes.data, pw.ctypes.data, ini.ctypes.data
        d.n_ref_features = last.n_features
        Tr = np.ascontiguousarray(last.Get_Pose(), np.float64).reshape(12).copy()
        Ts = np.ascontiguousarray(T_seed, np.float64).reshape(12).copy()
        d.T_ref_w, d.T_seed = Tr.ctypes.data, Ts.ctypes.data
        d.align = capi.AlignParams(*[int(v) for v in align])
        d.min_tracked = int(min_tracked)
        kfd = [capi.device_frame_of(ctx, k) for k in keyframes]
        kfh = (C.c_void_p * max(1, len(keyframes)))(*[k.handle for k in kfd])
        Tk = np.ascontiguousarray(np.array([k.Get_Pose() for k in keyframes], np.float64).reshape(len(keyframes), 12))
        d.kf, d.n_kf, d.T_kf_w = C.cast(kfh, C.c_void_p), len(keyframes), Tk.ctypes.data
        d.n_points = len(fm["found"])
        d.mp_world, d.mp_found, d.mp_bad, d.obs_offset = fm["pw"].ctypes.data, fm["found"].ctypes.data, fm["bad"].ctypes.data, fm["off"].ctypes.data
        d.obs_kf, d.obs_px, d.obs_level, d.obs_bearing = fm["okf"].ctypes.data, fm["opx"].ctypes.data, fm["olv"].ctypes.data, fm["ob"].ctypes.data
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            d.mask, d.mask_stride = mask.ctypes.data, mask.strides[0]
        d.cell_size, d.max_pyr_levels, d.max_matches, d.align2d_iters = cell_size, max_pyr_levels, int(max_matches), int(align2d_iters)
        d.pose_opt = capi.PoseOptParams(int(po_iterations), 0)
        self.desc = d
        self.res = capi.TrackResult()
        self.matches = np.zeros(max(1, int(max_matches)), capi.TRACK_MATCH_DTYPE)
        self.rn = np.zeros(max(1, int(max_matches)))
        self.cs = capi.camera_struct(cam)
        self._keep = (image, dref, px, bear, pw, ini, Tr, Ts, kfd, kfh, Tk, fm, mask)

    def run_raw(self) -> int:
        """The library call alone; returns its status (the new frame's handle is in self.res.frame)."""
        return self.ctx.lib.dsdtm_track_frame(self.ctx.handle, C.byref(self.cs), C.byref(self.desc), C.byref(self.res),
                                              self.matches.ctypes.data, self.rn.ctypes.data)

    def run(self) -> dict:
        self.ctx.check(self.run_raw())
        res = self.res
        sm = res.summary.as_dict()
        return dict(frame=capi.DeviceFrame(self.ctx, C.c_void_p(res.frame)), T_run=np.array(list(res.T_run)).reshape(3, 4),
                    n_tracked=int(res.n_tracked), lost=bool(res.lost), stats=res.stats.as_dict(), n_in_grid=int(res.n_in_grid),
                    replay_full_scan=bool(res.replay_full_scan), matches=self.matches[:res.n_matches].copy(),
                    T_opt=np.array(list(res.T_opt)).reshape(3, 4), summary=sm, residual_norm=self.rn[:sm["n_residual_blocks"]].copy())


def track_frame(ctx: capi.Context, cam, image, levels, last: Frame, T_seed, align, min_tracked, keyframes, map_points, **kw):
    """dsdtm_track_frame. `last` and the keyframes are Frames whose pyramids are (made) resident on the device; `align` =
    (max_level, min_level, max_iters, min_fts). Returns a dict: frame (capi.DeviceFrame of the new image), T_run, n_tracked, lost,
    stats, n_in_grid, matches (structured array: cell, point, px, level), T_opt, summary, residual_norm."""
    return TrackCall(ctx, cam, image, levels, last, T_seed, align, min_tracked, keyframes, map_points, **kw).run()


class Tracker:
    """Tracking's per-frame flow (src/Tracking.cpp:199-256) on top of track_frame, with the reference's side effects: the new
    Frame gets the pose, the features SearchLocalPoints creates (px, level, bearing, map point, mbInitial) and the refined
    pose; matched map points IncreaseFound (src/Feature_alignment.cpp:106); the mask gets its discs (:111); PoseOptimization's
    EraseFound walk runs on the block-ordered norms (src/Optimizer.cpp:80-92)."""

    def __init__(self, camera, ctx: capi.Context | None = None, max_level=None, min_level=None, max_iters=None, min_tracked=20):
        self.cam = camera
        self.ctx = ctx or capi.default_context()
        self.levels = int(Config.Get("Camera.MaxPyraLevels") if max_level is None else max_level)       # src/Tracking.cpp:20-24
        self.min_level = int(Config.Get("Camera.MinPyraLevels") if min_level is None else min_level)
        self.max_iters = int(Config.Get("Optimization.MaxIter") if max_iters is None else max_iters)
        self.min_tracked = int(min_tracked)
        self.last_result = None

    def TrackFrame(self, image, last: Frame, keyframes, map_points, img_mask=None):
        """Returns (cur Frame, n_tracked, matches as [(cell, MapPoint, px float32[2], level)])."""
        r = track_frame(self.ctx, self.cam, image, self.levels, last, last.Get_Pose(),
                        (self.levels, self.min_level, self.max_iters, int(Config.Get("Camera.Min_fts"))), self.min_tracked,
                        keyframes, map_points, mask=img_mask)
        self.last_result = r
        cur = Frame(self.cam, [np.ascontiguousarray(image, np.uint8)], r["T_run"])      # (level 0 only on the host: the pyramid is on the device)
        cur._device_frame = r["frame"]
        if r["lost"]:                                                                  # :208-214
            return cur, r["n_tracked"], []
        m = r["matches"]
        mps = [map_points[int(i)] for i in m["point"]]
        for mp in mps:
            mp.IncreaseFound()                                                         # src/Feature_alignment.cpp:106
        if img_mask is not None:
            for q in m["px"]:                                                          # :111 (cv::Point from Point2d rounds)
                search.fill_circle(img_mask, search.cvRound(float(q[0])), search.cvRound(float(q[1])), int(Config.Get("Camera.CellSize")), 0)
        if len(m):
            search.add_matched_features(cur, m["px"], m["level"], mps)                 # :108-114
        cur.Set_Pose(r["T_opt"])                                                       # src/Optimizer.cpp:78
        thresh = float(np.float32(Config.Get("Optimization.LocalBAthreshhold"))) / float(np.float32(self.cam.f))
        rn = r["residual_norm"]
        for i in range(len(rn)):                                                       # :80-92 (every feature has a block here: index = block)
            if rn[i] > thresh and not mps[i].IsBad():
                mps[i].EraseFound()
        return cur, r["n_tracked"], [(int(m["cell"][k]), mps[k], m["px"][k].copy(), int(m["level"][k])) for k in range(len(m))]
