"""GPU side of the feature detector: the HIP passes against the (reference-pinned) oracle."""
import ctypes as C

import numpy as np
import pytest

from dsdtm_amd import capi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ctx():
    return capi.default_context(0)


def gpu_fast10(ctx, img, barrier):
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    score, keep = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
    f = ctx.lib.dsdtm_debug_fast10
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, capi.u8p, C.c_int, C.c_int, C.c_int, C.c_int, capi.u8p, capi.u8p]
    ctx.check(f(ctx.handle, img.ctypes.data_as(capi.u8p), w, h, img.strides[0], barrier, score.ctypes.data_as(capi.u8p),
                keep.ctypes.data_as(capi.u8p)))
    ys, xs = np.nonzero(score)
    return np.stack([xs, ys, score[ys, xs], keep[ys, xs]], 1).astype(np.int32)


@pytest.mark.diag
def test_fast_maps_equal_the_reference_vectors(gpu_ctx_diag):
    """(The score map and the survivor map are intermediate results: only the diagnostic library has an entry that returns them,
    dsdtm_debug_fast10 — the same detect.hip, which DSDTM_DIAG does not touch. The release library's detector is held against the
    reference-pinned oracle through its public entries below.)
    Score map + non-max survivors of the device passes, corner by corner, against the vectors the
    reference's own FAST sources produced (tests/golden/fast_reference.npz)."""
    gpu_ctx = gpu_ctx_diag
    fx = np.load(H.golden_path("fast_reference.npz"))
    assert np.array_equal(gpu_fast10(gpu_ctx, fx["test1"], 75), fx["test1_b75"]) and len(fx["test1_b75"]) == 167
    for name in ("test1", "noise", "lowc", "tex", "narrow"):
        assert np.array_equal(gpu_fast10(gpu_ctx, fx[name], 20), fx[name + "_b20"]), name


@pytest.mark.diag
def test_fast_maps_equal_the_oracle_on_odd_shapes(gpu_ctx_diag, oracle):
    gpu_ctx = gpu_ctx_diag
    rng = np.random.default_rng(8)
    for shape in [(7, 22), (6, 40), (33, 257), (61, 300), (480, 640)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        img[img < 25] = 0; img[img > 230] = 255
        for barrier in (20, 3):
            assert np.array_equal(gpu_fast10(gpu_ctx, img, barrier), oracle.fast10_list(img, barrier)), (shape, barrier)
    big = rng.integers(0, 256, (50, 120), dtype=np.uint8)
    assert np.array_equal(gpu_fast10(gpu_ctx, big[:, 7:99], 20), oracle.fast10_list(big[:, 7:99], 20))     # strided input


def _cells(ctx, det, frame, thr):
    return det.detect_cells(frame, thr)


@pytest.mark.parametrize("size,levels", [((640, 480), 5), ((320, 240), 3), ((752, 480), 4)])
def test_detect_cells_equal_the_oracle(gpu_ctx, oracle, size, levels):
    from dsdtm_amd.feature_detection import Feature_detector
    from dsdtm_amd.frame import Config, Frame
    w, h = size
    if size == (752, 480):
        img = np.load(H.golden_path("fast_reference.npz"))["test1"]          # the reference's own test image
    else:
        img = np.clip(np.rint(synth.make_texture(h, w, w + levels)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(img, levels)
    old = Config.Get("Camera.MaxPyraLevels")
    Config.Set("Camera.MaxPyraLevels", levels)
    try:
        det = Feature_detector(w, h, ctx=gpu_ctx)
        fr = Frame(synth.Camera.tum(w, h), pyr)
        rng = np.random.default_rng(w)
        det.Set_ExistingFeatures(np.stack([rng.uniform(0, w - 1, 30), rng.uniform(0, h - 1, 30)], 1))
        occ = det.mvGrid_occupy.copy()
        for thr in (5.0, 40.0):
            got = _cells(gpu_ctx, det, fr, thr)
            want = oracle.detect_cells(pyr, levels, det.mCell_size, det.mGrid_cols, det.mGrid_rows, occ, thr)
            for g, wv, name in zip(got, want, ("score", "x", "y", "level")):
                assert np.array_equal(g, wv), (name, thr, np.nonzero(g != wv)[0][:5])
            assert (got[0][occ == 1] == np.float32(thr)).all() and (got[0] > thr).sum() > 20
        # a device-resident frame (pyramid built on the device from level 0) gives the same cells
        fr._device_frame = capi.DeviceFrame.from_image(gpu_ctx, img, levels)
        got_f = _cells(gpu_ctx, det, fr, 5.0)
        want = oracle.detect_cells(pyr, levels, det.mCell_size, det.mGrid_cols, det.mGrid_rows, occ, 5.0)
        assert all(np.array_equal(g, wv) for g, wv in zip(got_f, want))
        fr._device_frame.close()
    finally:
        Config.Set("Camera.MaxPyraLevels", old)


def test_detect_like_keyframe_creation(gpu_ctx, oracle):
    """Feature_detector::detect end to end (src/Tracking.cpp:416: detect(frame, 5.0) on a frame that
    already has tracked features) against the sequential restatement on oracle cells."""
    from dsdtm_amd.feature_detection import Feature_detector
    from dsdtm_amd.frame import Config, Frame
    from tests import detector_restatement as R
    img = np.clip(np.rint(synth.make_texture(480, 640, 99)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(img, 5)
    det = Feature_detector(640, 480, ctx=gpu_ctx)
    fr = Frame(synth.Camera.tum(640, 480), pyr)
    rng = np.random.default_rng(1)
    ex = np.stack([rng.uniform(20, 620, 60), rng.uniform(20, 460, 60)], 1).astype(np.float32)
    has = (rng.random(60) < 0.8).astype(np.uint8)
    fr.set_features(ex, np.zeros((60, 3)), np.zeros((60, 3)), has)
    det.Set_ExistingFeatures(ex)
    cells = oracle.detect_cells(pyr, 5, det.mCell_size, det.mGrid_cols, det.mGrid_rows, det.mvGrid_occupy.copy(), 5.0)
    want = R.detect(cells, 640, 480, det.mCell_size, det.mMax_fts, ex, has, int(Config.Get("Camera.Min_dist")))
    n_new = det.detect(fr, 5.0)
    got = [(int(fr.px[60 + i, 0]), int(fr.px[60 + i, 1]), int(fr.level[60 + i])) for i in range(n_new)]
    assert got == want and 20 < n_new <= det.mMax_fts - 60
    assert fr.n_features == 60 + n_new


def test_detect_honours_the_moving_object_mask(gpu_ctx):
    """Frame::Set_Mask (src/Frame.cpp:286-298) subtracts the thresholded moving-object mask from the image mask
    before detect() tests `mask != 255`: no new feature inside the masked region, the same features elsewhere
    (Max_fts lifted so that the comparison is not cut short)."""
    from dsdtm_amd.feature_detection import Feature_detector
    from dsdtm_amd.frame import Frame
    img = np.clip(np.rint(synth.make_texture(480, 640, 77)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(img, 5)
    ex = np.array([[100.0, 100.0]], np.float32)                  # one tracked feature: Set_Mask only runs with features
    out = []
    for masked in (False, True):
        det = Feature_detector(640, 480, ctx=gpu_ctx)
        det.mMax_fts = 10 ** 6
        fr = Frame(synth.Camera.tum(640, 480), pyr)
        fr.set_features(ex, np.zeros((1, 3)), np.zeros((1, 3)), np.ones(1, np.uint8))
        if masked:
            fr.mDynamicMask = np.zeros((480, 640), np.uint8)
            fr.mDynamicMask[100:300, 200:500] = 255
            fr.mDynamicMask[0:50, 0:50] = 150                    # below the threshold of 200: not masked
        det.detect(fr, 5.0)
        out.append({(int(x), int(y)) for x, y in fr.px[1:]})
    inside = {(x, y) for x, y in out[0] if 200 <= x < 500 and 100 <= y < 300}
    assert len(inside) > 10 and not any(200 <= x < 500 and 100 <= y < 300 for x, y in out[1])
    # every feature outside the region is still found; the only additions are corners next to the region that a
    # disc (radius CellSize) of a now-rejected corner used to suppress
    assert out[0] - inside <= out[1]
    extra = out[1] - out[0]
    assert all(200 - 26 <= x < 500 + 26 and 100 - 26 <= y < 300 + 26 for x, y in extra) and len(extra) < 20


@pytest.mark.parametrize("size,levels,n", [((640, 480), 5, 6), ((752, 480), 4, 3), ((640, 480), 5, 9), ((320, 240), 3, 12), ((752, 480), 3, 8),
                                           ((656, 490), 3, 8), ((640, 482), 3, 9)])
def test_batch_entry_equals_the_oracle_frame_by_frame(gpu_ctx, oracle, size, levels, n):
    """dsdtm_detect_cells_batch_device: n packed device pyramids in one call (strip kernel where the level rows are whole
    dwords, one thread per pixel where they are not: 752 -> 94 -> 47 columns), every frame with its own occupancy
    grid, against the oracle's cells frame by frame. From 8 frames the select pass is the batch kernel (4 x 4 pixels per
    thread, four survivors scored per round) when every level is whole dwords wide: 9 x 640x480x5, 12 x 320x240x3 and
    8 x 752x480x3 (the reference's test1.png among them) run it; 6 and 3 frames run the one-pixel-per-thread kernel.
    656x490 and 640x482 end the kernel's 4 x 4 strips inside the image: level heights 490 / 245 / 123 and 482 / 241 / 121, a
    level width (164) that is not a multiple of its 16-column thread groups."""
    import torch
    from dsdtm_amd.feature_detection import Feature_detector
    from dsdtm_amd.frame import Config
    w, h = size
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(w + n)
    imgs = [np.clip(np.rint(synth.make_texture(h, w, 300 + i)), 0, 255).astype(np.uint8) for i in range(n)]
    if size == (752, 480):
        imgs[0] = np.load(H.golden_path("fast_reference.npz"))["test1"]
        imgs[-1] = np.ascontiguousarray(imgs[0][::-1, ::-1])
    pyrs = [synth.build_pyramid(im, levels) for im in imgs]
    ws, hs, ss, offs, nbytes = capi.pyramid_layout(w, h, levels)
    pitch = (nbytes + 255) // 256 * 256
    packed = np.zeros((n, pitch), np.uint8)
    for i, p in enumerate(pyrs):
        for l in range(levels):
            packed[i, offs[l]:offs[l] + ws[l] * hs[l]] = p[l].reshape(-1)
    old = Config.Get("Camera.MaxPyraLevels")
    Config.Set("Camera.MaxPyraLevels", levels)
    try:
        det = Feature_detector(w, h, ctx=gpu_ctx)
    finally:
        Config.Set("Camera.MaxPyraLevels", old)
    G = det.mGrid_cols * det.mGrid_rows
    occ = (rng.random((n, G)) < 0.1).astype(np.uint8)
    prm = capi.DetectParams(det.mCell_size, det.mGrid_cols, det.mGrid_rows, levels, 20, 5.0)
    d_pyr, d_occ = torch.from_numpy(packed).to(dev), torch.from_numpy(occ).to(dev)
    d_score = torch.empty((n, pitch), dtype=torch.uint8, device=dev)
    d_key = torch.empty((n, G), dtype=torch.int64, device=dev)
    d_s = torch.empty((n, G), dtype=torch.float32, device=dev)
    d_x, d_y, d_l = (torch.empty((n, G), dtype=torch.int32, device=dev) for _ in range(3))
    wa, ha, sa = (C.c_int * levels)(*ws), (C.c_int * levels)(*hs), (C.c_int * levels)(*ss)
    oa = (C.c_size_t * levels)(*offs)
    for rep in range(2):                                        # the scratch is reusable from call to call
        gpu_ctx.check(gpu_ctx.lib.dsdtm_detect_cells_batch_device(
            gpu_ctx.handle, d_pyr.data_ptr(), pitch, n, levels, wa, ha, sa, oa, d_occ.data_ptr(), C.byref(prm), d_score.data_ptr(),
            d_key.data_ptr(), d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), d_l.data_ptr(), None))
        torch.cuda.synchronize()
    got = [t.cpu().numpy() for t in (d_s, d_x, d_y, d_l)]
    for i in range(n):
        want = oracle.detect_cells(pyrs[i], levels, det.mCell_size, det.mGrid_cols, det.mGrid_rows, occ[i], 5.0)
        for g, wv, name in zip(got, want, ("score", "x", "y", "level")):
            assert np.array_equal(g[i], wv), (i, name, np.nonzero(g[i] != wv)[0][:5])
        assert (got[0][i] > 5.0).sum() > 20
