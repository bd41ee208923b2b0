"""Host logic of the one-call tracked frame that needs no GPU: the flattening of the local map into the columns dsdtm_track_desc
takes (observations in the iteration order of mObservations, each carrying the observing feature's mpx / mlevel / mNormal), and
the descriptor the Python mirror builds from it."""
import ctypes as C

import numpy as np

from dsdtm_amd import capi, search, synth, tracking
from dsdtm_amd.frame import Frame


def _toy_map():
    cam = synth.Camera.tum(64, 48)
    kfs = []
    for k in range(3):
        kf = search.KeyFrame(cam, [np.zeros((48, 64), np.uint8)], np.eye(4)[:3], k)
        n = 4 + k
        px = np.arange(2 * n, dtype=np.float32).reshape(n, 2) + 10 * k
        kf.set_features(px, synth.bearing_from_px(cam, px), np.zeros((n, 3)), np.ones(n, np.uint8), level=np.arange(n, dtype=np.int32) % 3)
        kfs.append(kf)
    mps = [search.MapPoint(np.array([0.1, 0.2, 2.0]), {2: 5, 0: 1}, mnFound=4),          # dict order is NOT the iteration order
           search.MapPoint(np.array([0.3, -0.2, 3.0]), {}, mnFound=1, mbBad=True),
           search.MapPoint(np.array([-0.4, 0.0, 1.5]), {1: 0, 2: 3, 0: 3}, mnFound=-2)]
    return cam, kfs, mps


def test_flatten_local_map_orders_observations_like_get_closest_obs():
    cam, kfs, mps = _toy_map()
    fm = tracking.flatten_local_map(kfs, mps)
    assert list(fm["off"]) == [0, 2, 2, 5] and fm["off"].dtype == np.int32
    assert list(fm["okf"]) == [0, 2, 0, 1, 2]                                   # sorted keyframe index per point (search.get_closest_obs)
    assert np.array_equal(fm["pw"], np.array([m.mPose for m in mps])) and list(fm["found"]) == [4, 1, -2] and list(fm["bad"]) == [0, 1, 0]
    want = [(0, 1), (2, 5), (0, 3), (1, 0), (2, 3)]
    for j, (k, f) in enumerate(want):
        assert np.array_equal(fm["opx"][j], kfs[k].px[f]) and fm["olv"][j] == kfs[k].level[f] and np.array_equal(fm["ob"][j], kfs[k].bearing[f])
    assert fm["opx"].dtype == np.float32 and fm["ob"].dtype == np.float64 and fm["olv"].dtype == np.int32
    empty = tracking.flatten_local_map(kfs, [])
    assert list(empty["off"]) == [0] and len(empty["okf"]) == 0 and empty["opx"].shape == (0, 2) and empty["ob"].shape == (0, 3)


def test_track_match_dtype_mirrors_the_c_struct():
    assert capi.TRACK_MATCH_DTYPE.itemsize == C.sizeof(capi.TrackMatch) == 20
    assert capi.TRACK_MATCH_DTYPE.fields["px"][1] == capi.TrackMatch.px.offset == 8
    assert capi.TRACK_MATCH_DTYPE.fields["level"][1] == capi.TrackMatch.level.offset == 16
    assert C.sizeof(capi.TrackResult) % 8 == 0 and capi.TrackResult.T_opt.offset % 8 == 0
