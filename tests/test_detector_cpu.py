"""CPU side of the feature detector: the oracle's FAST restatement is PINNED — to
vectors produced by the reference's own vendored FAST sources (tests/golden/fast_reference.npz, incl.
the reference test's known answer, 167 corners on its test1.png at barrier 75) and, where
/root/reference exists, to a live build of those sources (oracle/_ref/libfast_ref.so)."""
import numpy as np
import pytest

from dsdtm_amd import synth
from tests import helpers as H


@pytest.fixture(scope="module")
def fixture():
    return np.load(H.golden_path("fast_reference.npz"))


def test_oracle_fast_reproduces_the_reference_vectors(oracle, fixture):
    assert len(fixture["test1_b75"]) == 167          # Thirdparty/fast/test/test.cpp:54 "BENCHMARK version extracted 167 features"
    assert np.array_equal(oracle.fast10_list(fixture["test1"], 75), fixture["test1_b75"])
    for name in ("test1", "noise", "lowc", "tex", "narrow"):
        got = oracle.fast10_list(fixture[name], 20)
        assert np.array_equal(got, fixture[name + "_b20"]), name
    assert fixture["narrow"].shape[1] < 22           # the plain-detector branch of fast_corner_detect_10_sse2


def test_oracle_fast_matches_a_live_build_of_the_reference_sources(oracle):
    if oracle.fast_ref_lib() is None:
        pytest.skip("/root/reference absent and no prebuilt oracle/_ref/libfast_ref.so")
    rng = np.random.default_rng(42)
    cases = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(50, 70), (33, 22), (7, 30), (90, 131), (6, 40)]]
    cases += [(100 + rng.integers(-40, 41, (64, 64))).astype(np.uint8)]
    cases += [np.clip(np.rint(synth.make_texture(96, 128, 3)), 0, 255).astype(np.uint8)]
    cases += [rng.integers(0, 256, (80, 200), dtype=np.uint8)[:, 3:163]]            # stride != width, unaligned base
    sat = rng.integers(0, 256, (40, 48), dtype=np.uint8); sat[sat < 30] = 0; sat[sat > 225] = 255   # saturating barriers
    cases.append(sat)
    for img in cases:
        for barrier in (20, 5, 75):
            want = oracle.fast10_list_reference(img, barrier)
            assert np.array_equal(oracle.fast10_list(img, barrier), want), (img.shape, barrier)


def test_shi_tomasi_is_the_smaller_eigenvalue(oracle):
    rng = np.random.default_rng(3)
    img = np.clip(np.rint(synth.make_texture(64, 80, 4)), 0, 255).astype(np.uint8)
    for _ in range(40):
        u, v = int(rng.integers(5, 75)), int(rng.integers(5, 59))
        f = img.astype(np.float64)
        box = np.s_[v - 4:v + 4, u - 4:u + 4]
        dx = (f[:, 2:] - f[:, :-2])[:, :][v - 4:v + 4, u - 5:u + 3]
        dy = (f[2:, :] - f[:-2, :])[v - 5:v + 3, u - 4:u + 4]
        M = np.array([[np.sum(dx * dx), np.sum(dx * dy)], [np.sum(dx * dy), np.sum(dy * dy)]]) / 128.0
        assert abs(oracle.shi_tomasi(img, u, v) - np.linalg.eigvalsh(M)[0]) < 1e-2 * max(1.0, M.trace())
    assert oracle.shi_tomasi(img, 4, 30) == 0.0 and oracle.shi_tomasi(img, 75, 30) == 0.0      # :173 too close to the boundary
    assert oracle.shi_tomasi(img, 30, 4) == 0.0 and oracle.shi_tomasi(img, 30, 59) == 0.0


def test_detect_cells_is_the_sequential_loop(oracle):
    """oracle_detect_cells against an independent Python loop over the corner lists (:94-107)."""
    tex = np.clip(np.rint(synth.make_texture(120, 160, 17)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 3)
    cell, cols, rows = 25, 7, 5
    occ = np.zeros(cols * rows, np.uint8); occ[[3, 11, 12]] = 1
    for thr in (5.0, 60.0):
        score, cx, cy, cl = oracle.detect_cells(pyr, 3, cell, cols, rows, occ, thr)
        best = [(np.float32(thr), 0, 0, 0)] * (cols * rows)
        for L in range(3):
            for x, y, s, keep in oracle.fast10_list(pyr[L], 20):
                if not keep:
                    continue
                k = ((y << L) // cell) * cols + (x << L) // cell
                if occ[k]:
                    continue
                sc = np.float32(oracle.shi_tomasi(pyr[L], int(x), int(y)))
                if sc > best[k][0]:
                    best[k] = (sc, int(x) << L, int(y) << L, L)
        assert [tuple(map(float, b)) for b in best] == [(float(score[k]), float(cx[k]), float(cy[k]), float(cl[k])) for k in range(cols * rows)]
        assert (score[occ == 1] == np.float32(thr)).all()
    assert (score > 60).sum() < cols * rows           # some cells have no corner above the higher threshold


def test_detect_bookkeeping_matches_the_sequential_restatement(oracle, monkeypatch):
    """Feature_detector.detect's host part (sort, mask discs, Max_fts cap) on oracle cells."""
    from dsdtm_amd.feature_detection import Feature_detector
    from dsdtm_amd.frame import Config, Frame
    from tests import detector_restatement as R
    tex = np.clip(np.rint(synth.make_texture(240, 320, 23)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 4)
    cam = synth.Camera.tum(320, 240)
    old = {k: Config.Get(k) for k in ("Camera.MaxPyraLevels", "Camera.Max_fts", "Camera.CellSize", "Camera.Min_dist")}
    try:
        for max_fts, n_existing in ((200, 0), (40, 12), (12, 12)):
            Config.Set("Camera.MaxPyraLevels", 4); Config.Set("Camera.Max_fts", max_fts)
            det = Feature_detector(320, 240)
            fr = Frame(cam, pyr)
            rng = np.random.default_rng(max_fts)
            ex = np.stack([rng.uniform(10, 310, n_existing), rng.uniform(10, 230, n_existing)], 1).astype(np.float32)
            has = (rng.random(n_existing) < 0.7).astype(np.uint8)
            fr.set_features(ex, np.zeros((n_existing, 3)), np.zeros((n_existing, 3)), has)
            det.Set_ExistingFeatures(ex)
            occ = det.mvGrid_occupy.copy()
            cells = oracle.detect_cells(pyr, 4, det.mCell_size, det.mGrid_cols, det.mGrid_rows, occ, 5.0)
            monkeypatch.setattr(det, "detect_cells", lambda frame, thr, c=cells: c)
            want = R.detect(cells, 320, 240, det.mCell_size, max_fts, ex, has, int(Config.Get("Camera.Min_dist")))
            n_new = det.detect(fr, 5.0)
            got = [(int(fr.px[n_existing + i, 0]), int(fr.px[n_existing + i, 1]), int(fr.level[n_existing + i])) for i in range(n_new)]
            assert got == want and n_new == len(want)
            assert fr.n_features <= max(max_fts, n_existing)
            assert (n_existing >= max_fts) or not det.mvGrid_occupy.any()      # :71-72 returns before ResetGrid (:152)
            if max_fts == 200:
                assert n_new > 30
    finally:
        for k, v in old.items():
            Config.Set(k, v)
