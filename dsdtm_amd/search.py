"""Feature_Alignment's reprojection grid and SearchLocalPoints (reference
src/Feature_alignment.cpp:22-126, FindMatchDirect :128-158) as a speculative GPU batch plus a
host replay of the reference's order-dependent side effects.

The reference walks the cells in index order and, per cell, tries candidates one by one
(warp + Align2D on the CPU) until one converges; a success paints a mask disc that can suppress
candidates of later cells, and the search stops after 200 matched cells. FindMatchDirect itself
has no side effect besides refining the candidate's pixel, so all candidates can be matched
speculatively in two launches (dsdtm_warp_patches, dsdtm_align2d_batch); the sequential rules are
then replayed on the (converged, px, level) triples. No image data is touched on the host.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from . import capi
from . import feature_alignment as FA
from .frame import Config, Frame

mHalf_PatchSize = 4


def cvRound(x: float) -> int:
    """OpenCV 2.4 cvRound (SSE2 cvtsd2si): round half to even."""
    return int(np.rint(x))


def is_in_image(cam, x: float, y: float, boundary: int, level: int = 0) -> bool:
    """Camera::IsInImage (src/Camera.cpp:187-193)."""
    return (cvRound(x) >= boundary and cvRound(x) < cam.width // (1 << level) - boundary and
            cvRound(y) >= boundary and cvRound(y) < cam.height // (1 << level) - boundary)


def _circle_spans(radius: int):
    """The horizontal spans (dy, half-width) cv::circle(.., -1) paints for `radius`: OpenCV 2.4 drawing.cpp Circle()
    (midpoint algorithm, filled by horizontal spans). A row can be painted by several spans: the widest one counts."""
    half = {}
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        for row, hw in ((-dy, dx), (dy, dx), (-dx, dy), (dx, dy)):
            half[row] = max(half.get(row, -1), hw)
        dy += 1
        err += plus
        plus += 2
        m = -1 if err > 0 else 0          # mask = (err <= 0) - 1
        err -= minus & m
        dx += m
        minus -= m & 2
    return half


_STENCILS = {}


def _circle_stencil(radius: int) -> np.ndarray:
    """The disc of _circle_spans as a (2r+1) x (2r+1) boolean stencil, built once per radius."""
    st = _STENCILS.get(radius)
    if st is None:
        st = np.zeros((2 * radius + 1, 2 * radius + 1), bool)
        for row, hw in _circle_spans(radius).items():
            st[radius + row, radius - hw:radius + hw + 1] = True
        _STENCILS[radius] = st
    return st


def fill_circle(mask: np.ndarray, cx: int, cy: int, radius: int, value: int = 0):
    """cv::circle(img, center, radius, value, -1) for 8-bit masks: OpenCV 2.4 drawing.cpp Circle()
    (midpoint algorithm, filled by horizontal spans, clipped to the image) — the spans are tabulated once per
    radius (_circle_stencil) and painted with one clipped slice assignment."""
    h, w = mask.shape
    st = _circle_stencil(radius)
    y0, y1 = max(cy - radius, 0), min(cy + radius, h - 1)
    x0, x1 = max(cx - radius, 0), min(cx + radius, w - 1)
    if y0 > y1 or x0 > x1:
        return
    sub = st[y0 - (cy - radius):y1 - (cy - radius) + 1, x0 - (cx - radius):x1 - (cx - radius) + 1]
    mask[y0:y1 + 1, x0:x1 + 1][sub] = value


@dataclasses.dataclass
class MapPoint:
    """What the search reads from DSDTM::MapPoint (include/MapPoint.h): position, observations
    (keyframe index -> feature index, iterated in keyframe-index order), found counter, bad flag."""
    mPose: np.ndarray
    mObservations: dict
    mnFound: int = 1
    mbBad: bool = False

    def Get_Pose(self):
        return self.mPose

    def IsBad(self):
        return self.mbBad

    def Get_FoundNums(self):
        return self.mnFound

    def IncreaseFound(self, n=1):
        self.mnFound += n

    def EraseFound(self, n=1):
        """MapPoint::EraseFound (src/MapPoint.cpp:183-198): at zero the point is flagged bad and loses its
        observations (SetBadFlag :91-109; the keyframe/map side of that is outside this path)."""
        self.mnFound -= n
        if self.mnFound <= 0:
            self.mbBad = True
            self.mObservations = {}


class KeyFrame(Frame):
    """Frame + the identity the observations refer to (include/Keyframe.h:78 shares mvImg_Pyr)."""

    def __init__(self, camera, img_pyr, T_c2w, kf_id):
        super().__init__(camera, img_pyr, T_c2w)
        self.mnId = kf_id


def get_closest_obs(mp: MapPoint, frame: Frame, keyframes):
    """MapPoint::Get_ClosetObs (src/MapPoint.cpp:133-174): observation whose viewing direction is
    closest to the frame's; rejected when cos < 0.5. Returns (kf_index, feature_index) or None."""
    if not mp.mObservations:
        return None
    v = frame.Get_CameraCnt() - mp.mPose
    v = v / np.linalg.norm(v)
    best, best_cos = None, 0.0
    first = None
    for kf_idx in sorted(mp.mObservations):
        if first is None:
            first = kf_idx
        r = keyframes[kf_idx].Get_CameraCnt() - mp.mPose
        r = r / np.linalg.norm(r)
        c = float(r @ v)
        if c > best_cos:
            best_cos, best = c, kf_idx
    if best is None:
        best = first
    if best_cos < 0.5:
        return None
    return best, mp.mObservations[best]


def bearing_of_pixel(cam, px) -> np.ndarray:
    """Frame::Add_Feature (src/Frame.cpp:83-92): mNormal = Camera::Pixel2Camera(mpx, 1.0).normalized(). The
    cv::Point2f overload (src/Camera.cpp:173-178) evaluates depth*(x - mcx)/mfx in FLOAT (all operands are
    float), the Eigen::Vector3d it returns is then normalised in double."""
    px = np.asarray(px, np.float32).reshape(-1, 2)
    one = np.float32(1.0)
    x = (one * (px[:, 0] - np.float32(cam.cx))) / np.float32(cam.fx)
    y = (one * (px[:, 1] - np.float32(cam.cy))) / np.float32(cam.fy)
    v = np.stack([x.astype(np.float64), y.astype(np.float64), np.ones(len(px))], axis=1)
    return v / np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])[:, None]


def add_matched_features(frame: Frame, px, level, map_points):
    """What ReprojectCell gives a matched candidate (src/Feature_alignment.cpp:108-114): a Feature at the
    refined pixel and search level with Mpt = the map point and mbInitial = true, its bearing, and the map
    point in the frame's mvMapPoints. p_world is the snapshot of Mpt->Get_Pose() that Sprase_ImgAlign::Run
    reads when this frame is the reference frame (src/Sprase_ImageAlign.cpp:93)."""
    new_px = np.asarray(px, np.float32).reshape(-1, 2)
    n_old = frame.n_features
    mpts = list(getattr(frame, "mvMapPoints", [None] * n_old))
    assert len(mpts) == n_old
    new = dict(px=new_px, level=np.asarray(level, np.int32), bearing=bearing_of_pixel(frame.mCamera, new_px),
               p_world=np.array([mp.Get_Pose() for mp in map_points], np.float64).reshape(-1, 3),
               initial=np.ones(len(new_px), np.uint8))
    for k, v in new.items():
        old = getattr(frame, k)
        setattr(frame, k, np.concatenate([old, v]) if n_old else v)
    frame.mvMapPoints = mpts + list(map_points)


class LocalPointSearch(FA.Feature_Alignment):
    """Feature_Alignment(CameraPtr) with ResetGrid / ReprojectPoint / SearchLocalPoints."""

    def __init__(self, camera, ctx=None, resident_frames: bool = False):
        super().__init__(camera, ctx)
        # resident_frames: keyframe and current pyramids stay on the device (dsdtm_frame) and every
        # candidate is matched by ONE library call (dsdtm_match_candidates_frames)
        self.resident_frames = bool(resident_frames)
        self.mMax_pts = Config.Get("Camera.Max_tkfts")
        self.mPyr_levels = Config.Get("Camera.MaxPyraLevels")
        self.mCell_size = Config.Get("Camera.CellSize")
        self.mGrid_Rows = int(np.ceil(camera.height / self.mCell_size))      # :29-30
        self.mGrid_Cols = int(np.ceil(camera.width / self.mCell_size))
        self.mCells = [[] for _ in range(self.mGrid_Rows * self.mGrid_Cols)]
        self.last_stats = {}

    def ResetGrid(self):                                                      # :46-52
        for c in self.mCells:
            c.clear()

    def ReprojectPoint(self, tFrame: Frame, tMPoint: MapPoint) -> bool:      # :54-69
        px = tFrame.World2Pixel(tMPoint.Get_Pose())
        if not (np.isfinite(px).all() and is_in_image(self.mCam, px[0], px[1], 8)):
            return False
        index = int(px[1] / self.mCell_size) * self.mGrid_Cols + int(px[0] / self.mCell_size)
        self.mCells[index].append([tMPoint, np.array(px, np.float64)])
        return True

    def SearchLocalPoints(self, tFrame: Frame, keyframes, img_mask: np.ndarray | None = None):
        """:71-121. tFrame gains features (px, level, map point); returns the list of
        (cell index, MapPoint, px, level) matches in the order the reference creates them."""
        cam = self.mCam
        if img_mask is None:
            img_mask = np.full((cam.height, cam.width), 255, np.uint8)       # Frame::mImgMask
        # ---- speculative part: every live candidate of every cell --------------------------------
        cand = []                                      # (cell, position in sorted cell, mp, px0)
        order = []
        for ci, cell in enumerate(self.mCells):
            cell.sort(key=lambda c: -c[0].Get_FoundNums())                   # :88, stable like std::list::sort
            for pos, (mp, px) in enumerate(cell):
                order.append((ci, pos))
                if mp.IsBad():
                    continue
                obs = get_closest_obs(mp, tFrame, keyframes)                 # :135
                if obs is None:
                    continue
                kf_idx, f_idx = obs
                kf = keyframes[kf_idx]
                rpx, rlv = kf.px[f_idx], int(kf.level[f_idx])
                if not is_in_image(cam, rpx[0] / (1 << rlv), rpx[1] / (1 << rlv), mHalf_PatchSize + 1, rlv):
                    continue                                                 # :138-140
                cand.append((ci, pos, kf_idx, f_idx))
        results = {}
        if cand:
            ck = np.array([c[2] for c in cand], np.int32)
            fi = [c[3] for c in cand]
            ref_px = np.array([keyframes[k].px[f] for k, f in zip(ck, fi)], np.float32)
            ref_lv = np.array([keyframes[k].level[f] for k, f in zip(ck, fi)], np.int32)
            ref_b = np.array([keyframes[k].bearing[f] for k, f in zip(ck, fi)], np.float64)
            # SolveAffineMatrix uses tReferFeature->Mpt->Get_Pose() (:167), i.e. the same map point
            pw = np.array([self.mCells[c[0]][c[1]][0].Get_Pose() for c in cand], np.float64)
            if self.resident_frames:
                cpx = np.array([self.mCells[c[0]][c[1]][1] for c in cand], np.float64)
                conv, pxl0, sl = FA.match_candidates_frames(tFrame, keyframes, cam, np.array([k.Get_Pose() for k in keyframes]),
                                                            tFrame.Get_Pose(), ck, ref_px, ref_lv, ref_b, pw, cpx,
                                                            self.mPyr_levels - 3, 10, ctx=self._ctx)
                for c, ok, p, s in zip(cand, conv, pxl0, sl):
                    results[(c[0], c[1])] = (bool(ok), p, int(s))
            else:
                aff, sl, pb, pp = FA.warp_patches([k.mvImg_Pyr for k in keyframes], cam,
                                                  np.array([k.Get_Pose() for k in keyframes]), tFrame.Get_Pose(),
                                                  ck, ref_px, ref_lv, ref_b, pw, self.mPyr_levels - 3, ctx=self._ctx)
                px0 = np.array([self.mCells[c[0]][c[1]][1] / (1 << int(s)) for c, s in zip(cand, sl)])   # :150
                conv, pxr = FA.align2d_batch(tFrame.mvImg_Pyr, pb, pp, sl, px0, 10, ctx=self._ctx)       # :152
                for c, ok, p, s in zip(cand, conv, pxr, sl):
                    results[(c[0], c[1])] = (bool(ok), p * (1 << int(s)), int(s))                        # :154-156
        # ---- replay of the sequential rules --------------------------------------------------------
        matches = []
        n_matches = 0
        for ci, cell in enumerate(self.mCells):                              # :75 index order
            for pos, (mp, px) in enumerate(cell):
                if mp.IsBad():                                               # :93
                    continue
                if img_mask[cvRound(px[1]), cvRound(px[0])] != 255:          # :96 (Point2f -> Point rounds)
                    continue
                r = results.get((ci, pos))
                if r is None or not r[0]:                                    # :101-104
                    continue
                ok, pxn, lvl = r
                mp.IncreaseFound()                                           # :106
                fill_circle(img_mask, cvRound(pxn[0]), cvRound(pxn[1]), self.mCell_size, 0)   # :111
                matches.append((ci, mp, pxn.astype(np.float32), lvl))        # Feature(px as Point2f, level)
                n_matches += 1
                break                                                        # :117 first success wins
            if n_matches >= 200:                                             # :80
                break
        # new Feature(frame, px, level); SetPose(map point) => Mpt, mbInitial = true (include/Feature.h:41-45);
        # Frame::Add_Feature (bearing, src/Frame.cpp:83-92) and Add_MapPoint (:113-114): the frame can now be
        # refined by Optimizer::PoseOptimization and serve as the reference frame of the next Run
        if matches:
            add_matched_features(tFrame, [m[2] for m in matches], [m[3] for m in matches], [m[1] for m in matches])
        self.last_stats = dict(candidates=len(cand), matched=n_matches)
        return matches
