"""Inputs built to EXERCISE every quirk of SURVEY.md §8.1, and the comparators the parity suite uses on them.

The reference holds no vectors for this path, so nothing external can show that the suite would notice a wrong
restatement of one of its quirks. These fixtures + oracle/mutants.h do it from the inside: each quirk has a mutant of the
oracle that "fixes" it, tests/test_mutants_cpu.py shows every mutant changes the outputs below by more than the
suite's tolerances, tests/test_mutants_gpu.py that the HIP path agrees with the faithful oracle and with no mutant.
The arrays are committed as tests/golden/quirks.npz (tests/golden/make_golden_quirks.py writes it from this module).
"""
from __future__ import annotations

import copy

import numpy as np

from dsdtm_amd import synth
from tests import helpers

ALIGN_PARAMS = (3, 0, 10)          # (max_level, min_level, max_iters) of every sparse case below
TOL = 10 * helpers.TIGHT_RAD       # the suite's FP64 pose tolerance (tests/test_sparse_align_gpu.py): 1e-8 rad / m


# ---------------------------------------------------------------------------------------------------------------
# sparse alignment cases: {name: (scene, params, min_fts, T_seed or None)}
# ---------------------------------------------------------------------------------------------------------------
def base_scene():
    """320x240, 3 levels, 160 features on a textured plane, with everything §8.1 singles out present at once:
    uninitialised features (Q3), map points that are exactly zero (Q3), features 3.1-3.6 px (x 2^level) from every border
    so that some cross the current-side border test during the iterations (Q3, Q10), map points that a bundle adjustment
    moved off their feature's ray (Q6: bearing * |P - C| != T_ref * P), f = 525 against fx = 517.3 (Q1)."""
    sc = copy.deepcopy(synth.make_scene(width=320, height=240, levels=3, n_patches=160, seed=0x51C, margin=14,
                                        xi=(0.012, -0.007, 0.005, 0.004, -0.003, 0.006),
                                        T_ref_w=synth.random_pose(np.random.default_rng(77), 0.4, 0.2)))
    rng = np.random.default_rng(0xB0D)
    W, H = 320, 240
    k = 0
    for lvl in range(3):                                       # four features per side and level, just inside the ref-side test
        s = 1 << lvl
        for d in (3.05, 3.2, 3.4, 3.6):
            sc.px[k] = (d * s, rng.uniform(40, 200)); k += 1
            sc.px[k] = (W - d * s, rng.uniform(40, 200)); k += 1
            sc.px[k] = (rng.uniform(40, 280), d * s); k += 1
            sc.px[k] = (rng.uniform(40, 280), H - d * s); k += 1
    sc.px = sc.px.astype(np.float32)
    sc.bearing = synth.bearing_from_px(sc.cam, sc.px)
    Xr = sc.bearing * (sc.depth / sc.bearing[:, 2:3])
    Rr, tr = sc.T_ref_w[:, :3], sc.T_ref_w[:, 3]
    # every third map point moved 1-3 cm off its ray (what LocalBundleAdjustment does to MapPoint::mPose)
    off = rng.normal(0, 0.02, Xr.shape)
    off[np.arange(len(Xr)) % 3 != 0] = 0.0
    sc.p_world = (Xr + off - tr) @ Rr
    sc.initial[:] = 1
    sc.initial[50::7] = 0
    sc.p_world[53::11] = 0.0
    return sc


def sparse_cases():
    base = base_scene()
    cases = {"main": (base, ALIGN_PARAMS, 15, None)}
    # a dark current frame: every interpolated intensity is exactly 0, the residual does not depend on the pose, chi2 repeats
    # EXACTLY from one iteration to the next — the only input on which `chi2New > chi2` (:328) and `>=` part ways
    # (interior features only: the visible set must not change while the pose drifts)
    dark = copy.copy(base)
    dark.cur_pyr = [np.zeros_like(a) for a in base.cur_pyr]
    inner = np.where((base.px[:, 0] > 60) & (base.px[:, 0] < 260) & (base.px[:, 1] > 60) & (base.px[:, 1] < 180))[0]
    dark.px, dark.bearing, dark.p_world, dark.initial = (a[inner].copy() for a in (base.px, base.bearing, base.p_world, base.initial))
    cases["dark"] = (dark, (3, 1, 4), 15, None)
    # the current camera seeded looking the other way (rotation by pi about the y axis of the reference camera): every point
    # has z < 0 and still projects into the image — the reference has no z > 0 test (:254-262)
    Ry = np.diag([-1.0, 1.0, -1.0])
    T4 = np.vstack([base.T_ref_w, [0, 0, 0, 1]])
    flip = np.eye(4)
    flip[:3, :3] = Ry
    cases["behind"] = (base, (3, 2, 2), 15, (flip @ T4)[:3].copy())
    # 20 features of which 8 are initialised, Min_fts = 15: Run counts ALL features (:34) and goes on
    few = copy.copy(base)
    sel = np.r_[60:80]
    few.px, few.bearing, few.p_world = base.px[sel].copy(), base.bearing[sel].copy(), base.p_world[sel].copy()
    few.initial = np.zeros(20, np.uint8)
    few.initial[:8] = 1
    few.p_world[few.p_world[:, 0] == 0.0] = base.p_world[0]
    cases["minfts"] = (few, ALIGN_PARAMS, 15, None)
    # the current camera seeded looking away (1.2 rad about y): no patch projects into the image — chi2 = 0/0 = NaN, H = 0, the
    # pseudo-inverse solve returns x = 0, the level ends on max|x| <= 1e-8 with the pose untouched (Q11)
    cases["away"] = (base, ALIGN_PARAMS, 15, (synth.se3_exp([0, 0, 0, 0, 1.2, 0]) @ T4)[:3].copy())
    return cases


def sparse_outputs(run, case):
    """run(scene, max_level, min_level, max_iters, min_fts=, T_seed=) -> (T, n, stats): the oracle wrapper or the GPU helper."""
    sc, prm, min_fts, T_seed = case
    T, n, st = run(sc, *prm, min_fts=min_fts, T_seed=T_seed)
    return dict(T=np.asarray(T, np.float64).copy(), n=int(n), iters=list(st["iters"]), exit_code=list(st["exit_code"]),
                n_ref=list(st["n_ref"]), n_vis=list(st["n_vis"]), chi2=[float(x) for x in st["chi2"]])


def sparse_first_difference(a, b, tol=TOL):
    """The first check of the suite's parity assertions (tests/test_sparse_align_gpu.py: pose, n_tracked, iterations, exit
    codes, n_ref / n_vis, chi2) that `b` fails against `a`; None when `b` passes them all."""
    if a["n"] != b["n"]:
        return "n_tracked"
    for k in ("iters", "exit_code", "n_ref", "n_vis"):
        if a[k] != b[k]:
            return k
    if not (np.all(np.isfinite(a["T"])) and np.all(np.isfinite(b["T"]))):
        if not np.array_equal(np.isfinite(a["T"]), np.isfinite(b["T"])):
            return "pose"
    else:
        ang, dt = synth.pose_error(a["T"], b["T"])
        if not (ang <= tol and dt <= tol):
            return "pose"
    ca, cb = np.array(a["chi2"]), np.array(b["chi2"])
    if not np.allclose(ca, cb, rtol=1e-9, atol=0, equal_nan=True):
        return "chi2"
    return None


# ---------------------------------------------------------------------------------------------------------------
# Align2D cases
# ---------------------------------------------------------------------------------------------------------------
def align2d_cases():
    """One 128x96 image, 3 levels. Rows: ordinary converging features; features under a brightness offset (the mean term, A4);
    features whose start lies too far to converge in 10 iterations or that leave the image (failure: px still written, A4);
    features whose window reaches u_r == cols - 4 / v_r == rows - 4 (admitted by the reference's bounds test, A3)."""
    rng = np.random.default_rng(0xA2D)
    tex = np.clip(np.rint(synth.make_texture(96, 128, 31)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 3)
    rows = []

    def add(level, centre, start, offset=0):
        img = pyr[level]
        pb, p = helpers.make_border_patches(img, [centre])
        pb = np.clip(pb[0].astype(np.int32) + offset, 0, 255).astype(np.uint8)
        rows.append((level, pb, pb.reshape(10, 10)[1:9, 1:9].reshape(64).copy(), np.array(start, np.float64)))

    for i in range(16):
        lv = i % 2
        h, w = pyr[lv].shape
        c = (rng.uniform(12, w - 12), rng.uniform(12, h - 12))
        add(lv, c, (c[0] + rng.uniform(-1.2, 1.2), c[1] + rng.uniform(-1.2, 1.2)))
    for i in range(6):                                         # brightness offset +-25 grey levels
        c = (rng.uniform(20, 108), rng.uniform(20, 76))
        add(0, c, (c[0] + rng.uniform(-1.0, 1.0), c[1] + rng.uniform(-1.0, 1.0)), offset=25 if i % 2 else -25)
    for i in range(6):                                         # hopeless starts: 6-9 px away
        c = (rng.uniform(30, 98), rng.uniform(30, 66))
        a = rng.uniform(0, 2 * np.pi)
        add(0, c, (c[0] + 7.5 * np.cos(a), c[1] + 7.5 * np.sin(a)))
    h, w = pyr[0].shape
    # two constant reference patches: H singular, H.inverse() is all NaN / inf, the first update makes u NaN (A2: no check)
    for i in range(2):
        c = (40.0 + 20 * i, 40.0)
        add(0, c, (c[0] + 0.4, c[1] - 0.3))
        lv, pb, p_, st = rows[-1]
        rows[-1] = (lv, np.full(100, 90 + 40 * i, np.uint8), np.full(64, 90 + 40 * i, np.uint8), st)
    for i in range(6):                                         # the admitted last column / row
        if i % 2:
            c = (w - 4 + 0.3 + 0.1 * i, rng.uniform(20, 70))
        else:
            c = (rng.uniform(20, 100), h - 4 + 0.2 + 0.1 * i)
        add(0, (min(c[0], w - 6.0), min(c[1], h - 6.0)), c)
    level = np.array([r[0] for r in rows], np.int32)
    return dict(pyr=pyr, level=level, patch_border=np.array([r[1] for r in rows]), patch=np.array([r[2] for r in rows]),
                px0=np.array([r[3] for r in rows]))


def align2d_first_difference(a, b):
    """a, b = (converged flags, pixels): flags and pixels must be bit-identical (tests/test_align2d_gpu.py)."""
    if not np.array_equal(a[0], b[0]):
        return "converged"
    if not np.array_equal(a[1], b[1], equal_nan=True):
        return "px"
    return None


# ---------------------------------------------------------------------------------------------------------------
# warp prelude cases (SolveAffineMatrix / GetBestSearchLevel / WarpAffine / GetPatchNoBoarder)
# ---------------------------------------------------------------------------------------------------------------
def warp_cases():
    """Candidates seen from a current camera at 1x, 0.62x and 0.45x of the keyframe's distance to the point: det(A) ~ 1, ~2.6
    (between the mutant's threshold 2 and the reference's 3) and ~4.9 (search level 1, where the integer division W1 collapses
    the warp); reference features on levels 0 and 1 (W2)."""
    rng = np.random.default_rng(0x3A9)
    tex = np.clip(np.rint(synth.make_texture(96, 128, 33)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 3)
    cam = synth.Camera.tum(128, 96)
    m = 36
    T_kf = np.array([np.eye(4)[:3], synth.random_pose(rng, 0.05, 0.03)])
    ck = (np.arange(m) % 2).astype(np.int32)
    rl = ((np.arange(m) // 2) % 2).astype(np.int32)
    rp = np.stack([rng.uniform(24, 104, m), rng.uniform(24, 72, m)], 1).astype(np.float32)
    rb = synth.bearing_from_px(cam, rp)
    depth = rng.uniform(1.5, 2.5, m)
    pw = np.array([T_kf[ck[i]][:, :3].T @ (rb[i] * depth[i] - T_kf[ck[i]][:, 3]) for i in range(m)])
    # one current pose per group of 12 candidates: the camera advanced along its optical axis
    groups = []
    for g, frac in enumerate((1.0, 0.62, 0.45)):
        T = np.eye(4)
        T[:3] = synth.random_pose(rng, 0.02, 0.02)
        T[2, 3] -= (1.0 - frac) * 2.0
        groups.append((T[:3].copy(), np.arange(g * 12, (g + 1) * 12)))
    return dict(pyr=pyr, cam=cam, T_kf=T_kf, groups=groups, cand_kf=ck, ref_level=rl, ref_px=rp, ref_bearing=rb, p_world=pw,
                max_search_level=2)


def warp_outputs(fn, w):
    """fn(kf_pyrs, cam, T_kf_w, T_cur_w, cand_kf, ref_px, ref_level, ref_bearing, p_world, max_search_level) ->
    (affine, search_level, patch_border, patch), per group of the fixture; concatenated."""
    outs = []
    for T_cur, idx in w["groups"]:
        outs.append(fn([w["pyr"], w["pyr"]], w["cam"], w["T_kf"], T_cur, w["cand_kf"][idx], w["ref_px"][idx], w["ref_level"][idx],
                       w["ref_bearing"][idx], w["p_world"][idx], w["max_search_level"]))
    return tuple(np.concatenate([o[i] for o in outs]) for i in range(4))


def warp_first_difference(a, b):
    """affine bit-identical, levels equal, patch bytes equal (tests/test_search_gpu.py::test_warp_patches_match_oracle)."""
    if not np.array_equal(a[1], b[1]):
        return "search_level"
    if not np.array_equal(a[0], b[0]):
        return "affine"
    if not np.array_equal(a[2], b[2]):
        return "patch_border bytes"
    if not np.array_equal(a[3], b[3]):
        return "patch bytes"
    return None


# ---------------------------------------------------------------------------------------------------------------
# SearchLocalPoints worlds (tests/test_search_gpu.py::make_world): the standard one and two dense ones whose matches hit the
# 200-cell cap, with features down to 3 px from their level's border and 15-px cells
# ---------------------------------------------------------------------------------------------------------------
SEARCH_WORLDS = {
    "std": dict(seed=3, n_points=900, cell=25),
    "dense": dict(seed=5, n_points=2000, cell=15, obs_margin=3),
    "dense_border": dict(seed=5, n_points=2000, cell=15, obs_margin=3, uv_margin=4),
}


def search_world(name):
    """(cam, keyframes, current frame, map points, cell size) of a named world; the grid is filled by the caller."""
    from dsdtm_amd.frame import Config
    from tests.test_search_gpu import make_world
    kw = dict(SEARCH_WORLDS[name])
    Config.Set("Camera.CellSize", kw["cell"])
    Config.Set("Camera.MaxPyraLevels", 5)
    cam, kfs, cur, mps = make_world(kw.pop("seed"), **kw)
    return cam, kfs, cur, mps, kw["cell"]


def search_restated(name, mutant=None):
    """The sequential CPU restatement (tests/search_restatement.py) on a named world: ([(cell, map point index, px, level)], mask)."""
    from dsdtm_amd import search
    from tests import search_restatement as SR
    cam, kfs, cur, mps, cell = search_world(name)
    s = search.LocalPointSearch.__new__(search.LocalPointSearch)         # the grid bookkeeping only: no GPU context
    search.FA.Feature_Alignment.__init__(s, cam, None)
    s.mCell_size = cell
    s.mGrid_Rows, s.mGrid_Cols = int(np.ceil(cam.height / cell)), int(np.ceil(cam.width / cell))
    s.mCells = [[] for _ in range(s.mGrid_Rows * s.mGrid_Cols)]
    for mp in mps:
        s.ReprojectPoint(cur, mp)
    mask = np.full((cam.height, cam.width), 255, np.uint8)
    idx = {id(mp): i for i, mp in enumerate(mps)}
    out = SR.search_local_points(s.mCells, cur, kfs, cam, cell, 5, mask, mutant=mutant)
    return [(int(o[0]), idx[id(o[1])], float(o[2][0]), float(o[2][1]), int(o[3])) for o in out], mask


def search_first_difference(a, b):
    """a, b = (match list, mask): same cells in the same order, same map point, level and refined pixel per cell, same mask
    (tests/test_search_gpu.py::test_search_local_points_matches_sequential_reference_flow)."""
    (la, ma), (lb, mb) = a, b
    if len(la) != len(lb):
        return "match count"
    if [x[0] for x in la] != [x[0] for x in lb]:
        return "cells"
    if [x[1] for x in la] != [x[1] for x in lb]:
        return "map points"
    if [x[4] for x in la] != [x[4] for x in lb]:
        return "search level"
    if [x[2:4] for x in la] != [x[2:4] for x in lb]:
        return "px"
    if not np.array_equal(ma, mb):
        return "mask"
    return None


# ---------------------------------------------------------------------------------------------------------------
# pose-only refinement problems (Optimizer::PoseOptimization): observations on levels 0..3 (the residual is divided by
# 1 << level, the Jacobian is not), outliers (the Cauchy loss), unused features (Mpt / IsBad / mbInitial), a seed far enough
# from the optimum for tens of iterations
# ---------------------------------------------------------------------------------------------------------------
POSE_PROBLEMS = [dict(seed=101, n=200, max_level=3, outlier_frac=0.15, unused_frac=0.2),
                 dict(seed=102, n=120, max_level=3, seed_t=0.12, seed_w=0.1, unused_frac=0.1),
                 dict(seed=103, n=300, max_level=2, outlier_frac=0.3, noise_px=1.0, unused_frac=0.25)]


def pose_problems():
    return [synth.make_pose_problem(**kw) for kw in POSE_PROBLEMS]


def pose_first_difference(a, b, tol=1e-10):
    """a, b = lists of (T, residual norms, summary): the assertions of tests/test_pose_opt_gpu.py::assert_same."""
    for (Ta, ra, sa), (Tb, rb, sb) in zip(a, b):
        for k in ("n_residual_blocks", "iterations", "successful_steps", "termination"):
            if sa[k] != sb[k]:
                return k
        ang, dt = synth.pose_error(Ta, Tb)
        if not (ang <= tol and dt <= tol):
            return "pose"
        if len(ra) != len(rb) or not np.allclose(ra, rb, rtol=0, atol=tol):
            return "residual norms"
        if not np.allclose([sa["initial_cost"], sa["final_cost"]], [sb["initial_cost"], sb["final_cost"]], rtol=1e-10, atol=1e-300):
            return "cost"
    return None


# ---------------------------------------------------------------------------------------------------------------
# pyramid + detector: a textured 160x120 frame whose right half repeats a 12x12 tile (identical corners with identical
# Shi-Tomasi scores inside one 25-px cell: the strict `>` of Feature_detection.cpp:104 keeps the FIRST of them)
# ---------------------------------------------------------------------------------------------------------------
def detector_image():
    img = np.clip(np.rint(synth.make_texture(120, 160, 77)), 0, 255).astype(np.uint8)
    tile = img[20:32, 30:42].copy()
    img[:, 88:] = np.tile(tile, (10, 6))
    return img


def detector_outputs(pyrdown, fast10_list, detect_cells):
    """pyrdown(img) -> next level; fast10_list(img, barrier) -> (x, y, score, is_nonmax) rows; detect_cells(pyr, levels, cell, cols,
    rows, occupied, threshold) -> (score, x, y, level) per cell. CPU oracle wrappers or their GPU twins."""
    img = detector_image()
    l1 = pyrdown(img)
    l2 = pyrdown(l1)
    pyr = [img, l1, l2]
    return dict(pyr1=l1, pyr2=l2, fast=fast10_list(img, 20), cells=detect_cells(pyr, 3, 25, 7, 5, None, 5.0))


def detector_first_difference(a, b):
    if not (np.array_equal(a["pyr1"], b["pyr1"]) and np.array_equal(a["pyr2"], b["pyr2"])):
        return "pyramid bytes"
    if not np.array_equal(a["fast"], b["fast"]):
        return "fast score map / survivors"
    for x, y, name in zip(a["cells"], b["cells"], ("cell score", "cell x", "cell y", "cell level")):
        if not np.array_equal(x, y):
            return name
    return None
