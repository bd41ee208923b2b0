"""Mutation sensitivity of the parity suite, GPU half.

On the quirk fixtures (tests/golden/quirks.npz, tests/quirk_fixtures.py) the HIP path, called through the C ABI, must
  * agree with the FAITHFUL oracle under the suite's own assertions (pose 1e-8, identical n_tracked / iterations / exit codes /
    n_ref / n_vis; Align2D flags and pixels, affine, search levels and patch bytes bit for bit; identical match lists and masks), and
  * DISAGREE with every mutant of the oracle — each quirk of SURVEY.md §8.1 "fixed" — at the case and assertion
    tests/mutant_runs.py::TABLE names.
So each quirk is exercised rather than assumed: a kernel (or an oracle) that got one wrong could not pass both halves."""
import numpy as np
import pytest

from dsdtm_amd import feature_alignment as FA
from dsdtm_amd import search
from dsdtm_amd.optimizer import pose_optimization
from tests import helpers as H
from tests import mutant_runs as M
from tests import quirk_fixtures as Q

pytestmark = pytest.mark.gpu


def gpu_pyrdown(ctx, img):
    """One level of Frame::ComputeImagePyramid through the C ABI (dsdtm_pyrdown)."""
    import ctypes as C
    from dsdtm_amd import capi
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    outs = (C.c_void_p * 2)(None, out.ctypes.data)
    strides = (C.c_int * 2)(w, out.strides[0])
    ctx.check(ctx.lib.dsdtm_pyrdown(ctx.handle, img.ctypes.data_as(capi.u8p), w, h, img.strides[0], 2, outs, strides))
    return out


def gpu_detect_cells(ctx, pyr, levels, cell, cols, rows, occupied, thr):
    """The per-cell part of Feature_detector::detect through the reference-shaped class."""
    from dsdtm_amd import synth
    from dsdtm_amd.feature_detection import Feature_detector
    from dsdtm_amd.frame import Config, Frame
    h, w = pyr[0].shape
    old, oldc = Config.Get("Camera.MaxPyraLevels"), Config.Get("Camera.CellSize")
    Config.Set("Camera.MaxPyraLevels", levels)
    Config.Set("Camera.CellSize", cell)
    try:
        det = Feature_detector(w, h, ctx=ctx)
        assert (det.mGrid_cols, det.mGrid_rows) == (cols, rows)
        return det.detect_cells(Frame(synth.Camera.tum(w, h), pyr), thr)
    finally:
        Config.Set("Camera.MaxPyraLevels", old)
        Config.Set("Camera.CellSize", oldc)


def gpu_search(ctx, name):
    cam, kfs, cur, mps, cell = Q.search_world(name)
    s = search.LocalPointSearch(cam, ctx=ctx)
    s.ResetGrid()
    for mp in mps:
        s.ReprojectPoint(cur, mp)
    mask = np.full((cam.height, cam.width), 255, np.uint8)
    idx = {id(mp): i for i, mp in enumerate(mps)}
    got = s.SearchLocalPoints(cur, kfs, mask)
    return [(int(g[0]), idx[id(g[1])], float(g[2][0]), float(g[2][1]), int(g[3])) for g in got], mask


@pytest.fixture(scope="module")
def hip(gpu_ctx):
    """The product path's outputs for every fixture case."""
    fx = M.fixtures()
    out = {}
    for name, case in fx["sparse"].items():
        out["sparse:" + name] = Q.sparse_outputs(lambda sc, *p, **kw: H.gpu_sparse_align(sc, *p, ctx=gpu_ctx, **kw), case)
    a = fx["align2d"]
    out["align2d"] = FA.align2d_batch(a["pyr"], a["patch_border"], a["patch"], a["level"], a["px0"], 10, ctx=gpu_ctx)
    out["warp"] = Q.warp_outputs(lambda *args: FA.warp_patches(*args, ctx=gpu_ctx), fx["warp"])
    for name in Q.SEARCH_WORLDS:
        out["search:" + name] = gpu_search(gpu_ctx, name)
    from tests.test_detector_gpu import gpu_fast10
    from dsdtm_amd import capi
    diag_ctx = capi.default_context(0, diag=True)      # the FAST score / survivor maps are intermediate results: only the
    out["detector"] = Q.detector_outputs(lambda im: gpu_pyrdown(gpu_ctx, im), lambda im, b: gpu_fast10(diag_ctx, im, b),   # diagnostic library returns them (same detect.hip)
                                         lambda *a: gpu_detect_cells(gpu_ctx, *a))
    out["pose_opt"] = []
    for P in fx["pose_opt"]:
        T = np.ascontiguousarray(P.T_seed, np.float64).reshape(12).copy()
        rn, sm = pose_optimization(gpu_ctx, P.bearing, P.p_world, P.level, P.use, T)
        out["pose_opt"].append((T.reshape(3, 4), rn, sm))
    return out


@pytest.fixture(scope="module")
def faithful():
    return M.cpu_outputs()


def test_hip_path_agrees_with_the_faithful_oracle_on_every_quirk_fixture(hip, faithful):
    diffs = M.first_differences(faithful, hip)
    assert set(diffs) == set(faithful) and all(v is None for v in diffs.values()), diffs


def test_hip_path_agrees_with_the_committed_outputs(hip):
    g = np.load(H.golden_path("quirks.npz"))
    for name in ("main", "dark", "behind", "minfts", "away"):
        o = hip["sparse:" + name]
        want = dict(T=g[f"{name}_out_T"], n=int(g[f"{name}_out_n"]), chi2=list(g[f"{name}_out_chi2"]),
                    **{k: list(g[f"{name}_out_{k}"]) for k in ("iters", "exit_code", "n_ref", "n_vis")})
        assert Q.sparse_first_difference(want, o) is None, name
    assert np.array_equal(hip["align2d"][0], g["a2d_out_conv"]) and np.array_equal(hip["align2d"][1], g["a2d_out_px"], equal_nan=True)
    for i, k in enumerate(("warp_out_affine", "warp_out_level", "warp_out_border", "warp_out_patch")):
        assert np.array_equal(hip["warp"][i], g[k]), k
    for name in Q.SEARCH_WORLDS:
        lst, mask = hip["search:" + name]
        assert np.array_equal(np.array(lst, np.float64).reshape(-1, 5), g[f"search_{name}_matches"]), name
        assert np.array_equal(np.packbits(mask == 255, axis=1), g[f"search_{name}_mask_rows"]), name


@pytest.mark.parametrize("mutant", M.ALL_MUTANTS)
def test_hip_path_disagrees_with_the_mutant(hip, mutant):
    case, check, quirk, cite = M.TABLE[mutant]
    dom = M.domain(mutant)
    out = M.cpu_outputs(mutant, domains=(dom,), search_worlds=[case.split(":")[1]] if dom == "search" else None)
    got = M.first_difference(case, out[case], hip[case])
    assert got == check, f"{mutant} ({quirk}, {cite}): the suite's assertions on {case} give {got!r} for the HIP path against this mutant, expected {check!r}"
