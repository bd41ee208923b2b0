"""Sequential CPU restatement of SearchLocalPoints / ReprojectCell / FindMatchDirect
(reference src/Feature_alignment.cpp:71-158) on top of the C oracle's warp + Align2D: one
candidate at a time, mask test BEFORE matching, exactly as the reference loops. Written
separately from dsdtm_amd/search.py (which matches speculatively and replays) — the two must agree."""
import numpy as np

from tests import oracle_lib


def cv_round(x):
    return int(np.rint(x))


def in_image(cam, x, y, b, lvl=0):
    return (cv_round(x) >= b and cv_round(x) < cam.width // (1 << lvl) - b and
            cv_round(y) >= b and cv_round(y) < cam.height // (1 << lvl) - b)


def circle_filled(mask, cx, cy, r):
    """OpenCV 2.4 Circle() fill, written independently as the union of the 4 span families."""
    h, w = mask.shape
    err, dx, dy, plus, minus = 0, r, 0, 1, 2 * r - 1
    spans = []
    while dx >= dy:
        spans += [(cy - dy, cx - dx, cx + dx), (cy + dy, cx - dx, cx + dx), (cy - dx, cx - dy, cx + dy), (cy + dx, cx - dy, cx + dy)]
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2
    for y, x1, x2 in spans:
        if 0 <= y < h and max(x1, 0) <= min(x2, w - 1):
            mask[y, max(x1, 0):min(x2, w - 1) + 1] = 0


# The order-dependent rules of SURVEY.md §8.1 S1 / W3 "fixed" one at a time (tests/test_mutants_*.py: the parity suite must
# tell each of these from the faithful flow)
SEARCH_MUTANTS = {
    "S1_SHUFFLE": "cells visited in the shuffled mCellOrder (:38-43) instead of index order (:75)",
    "S1_NOCAP": "no stop after 200 matched cells (:80)",
    "S1_NOSORT": "candidates of a cell not sorted by Get_FoundNums() (:88,123-126)",
    "S1_NOMASK": "the mask test before matching dropped (:96)",
    "S1_NOBAD": "bad map points not skipped (:93)",
    "S1_ALLCANDS": "no break after the first success in a cell (:115)",
    "W3_BORDER": "reference pixel kept 8 px (the grid's border, :62) instead of 5 px inside its level image (:138-140)",
    "W3_FIRSTOBS": "the first observation instead of the closest view (src/MapPoint.cpp:148-171)",
}


def search_local_points(cells, frame, keyframes, cam, cell_size, max_levels, mask, mutant=None):
    """cells: list of lists of [MapPoint, px]. Returns [(cell, mp, px float32, level)]."""
    assert mutant is None or mutant in SEARCH_MUTANTS, mutant
    out = []
    matches = 0
    order = list(range(len(cells)))
    if mutant == "S1_SHUFFLE":
        np.random.default_rng(0).shuffle(order)
    for ci in order:
        cell = cells[ci]
        if mutant != "S1_NOSORT":
            cell.sort(key=lambda c: -c[0].mnFound)
        for cand in cell:
            mp, px = cand
            if mp.mbBad and mutant != "S1_NOBAD":
                continue
            if mask[cv_round(px[1]), cv_round(px[0])] != 255 and mutant != "S1_NOMASK":
                continue
            # --- FindMatchDirect ---
            if not mp.mObservations:
                continue
            v = frame.Get_CameraCnt() - mp.mPose
            v /= np.linalg.norm(v)
            best, best_cos = None, 0.0
            for k in sorted(mp.mObservations):
                r = keyframes[k].Get_CameraCnt() - mp.mPose
                r /= np.linalg.norm(r)
                if float(r @ v) > best_cos:
                    best_cos, best = float(r @ v), k
                    if mutant == "W3_FIRSTOBS":
                        break
            if best is None or best_cos < 0.5:
                continue
            kf = keyframes[best]
            f = mp.mObservations[best]
            rpx, rlv = kf.px[f], int(kf.level[f])
            if not in_image(cam, rpx[0] / (1 << rlv), rpx[1] / (1 << rlv), 8 if mutant == "W3_BORDER" else 5, rlv):
                continue
            aff, sl, pb, pp = oracle_lib.warp_patches([kf.mvImg_Pyr], cam, [kf.Get_Pose()], frame.Get_Pose(), [0],
                                                      [rpx], [rlv], [kf.bearing[f]], [mp.mPose], max_levels - 3)
            lvl = int(sl[0])
            img = frame.mvImg_Pyr[lvl]
            ok, pxn = oracle_lib.align2d(img, pb[0], pp[0], 10, px / (1 << lvl))
            cand[1] = pxn * (1 << lvl)                       # tPt is written back (:154)
            if not ok:
                continue
            mp.mnFound += 1
            circle_filled(mask, cv_round(cand[1][0]), cv_round(cand[1][1]), cell_size)
            out.append((ci, mp, cand[1].astype(np.float32), lvl))
            matches += 1
            if mutant != "S1_ALLCANDS":
                break
        if matches >= 200 and mutant != "S1_NOCAP":
            break
    return out
