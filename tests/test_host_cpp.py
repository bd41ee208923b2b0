"""The C++ host layer (dsdtm_amd/host/dsdtm_host.hpp: reference class names over the C ABI) builds
with g++ against libdsdtm_amd.so and, on a GPU, reproduces the oracle through a driver shaped like
the reference's Test/test_SpraseImg_alignment.cpp and Test/test_Feature_alignment.cpp."""
import os
import struct
import subprocess

import numpy as np
import pytest

from dsdtm_amd import capi, synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dsdtm_amd", "host")
EXE = os.path.join(HOST, "example_align")


def build_example(name="example_align"):
    exe = os.path.join(HOST, name)
    src = os.path.join(HOST, name + ".cpp")
    hdr = os.path.join(HOST, "dsdtm_host.hpp")
    lib = capi.lib_path()
    if (not os.path.exists(exe)) or any(os.path.getmtime(p) > os.path.getmtime(exe) for p in (src, hdr, lib)):
        subprocess.run(["g++", "-O2", "-std=c++14", "-Wall", "-o", exe, src, lib,
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"],
                       check=True)
    return exe


def dump_scene(path, sc, params, min_fts, border, patch, px0):
    with open(path, "wb") as f:
        L = len(sc.ref_pyr)
        f.write(struct.pack("<8i", L, len(sc.px), params[0], params[1], params[2], min_fts, sc.cam.width, sc.cam.height))
        f.write(struct.pack("<5f", sc.cam.fx, sc.cam.fy, sc.cam.cx, sc.cam.cy, sc.cam.f))
        for pyr in (sc.ref_pyr, sc.cur_pyr):
            for l in pyr:
                f.write(np.ascontiguousarray(l).tobytes())
        for i in range(len(sc.px)):
            f.write(sc.px[i].astype("<f4").tobytes() + sc.bearing[i].astype("<f8").tobytes() +
                    sc.p_world[i].astype("<f8").tobytes() + bytes([int(sc.initial[i])]))
        f.write(np.ascontiguousarray(sc.T_ref_w, "<f8").tobytes() + np.ascontiguousarray(sc.T_cur_w_seed, "<f8").tobytes())
        f.write(bytes(border) + bytes(patch) + np.asarray(px0, "<f8").tobytes())


def test_cpp_host_layer_builds_against_the_c_abi():
    assert all(os.path.exists(build_example(n)) for n in ("example_align", "example_search", "example_pose_opt", "example_rgbd",
                                                          "example_track", "example_batch"))


@pytest.mark.gpu
def test_cpp_driver_matches_oracle(tmp_path, oracle):
    exe = build_example()
    sc = synth.make_scene(width=320, height=240, levels=3, n_patches=140, seed=77, margin=12, frac_uninitial=0.05)
    img = sc.cur_pyr[0]
    pb, p = H.make_border_patches(img, [(150.3, 101.6)])
    px0 = np.array([150.3 + 0.9, 101.6 - 0.7])
    scene = tmp_path / "scene.bin"
    dump_scene(scene, sc, (3, 0, 10), 15, pb[0], p[0], px0)
    out = subprocess.run([exe, str(scene)], capture_output=True, text=True, check=True).stdout.split("\n")
    n = int(out[0].split()[1])
    T = np.array([float(v) for v in out[1].split()[1:]]).reshape(3, 4)
    iters = [int(v) for v in out[2].split()[1:]]
    a2d = out[3].split()
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    H.assert_pose_close(T, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what="c++ driver")
    assert n == no and iters == so["iters"][:3]
    oko, pxo = oracle.align2d(img, pb[0], p[0], 10, px0)
    assert bool(int(a2d[1])) == oko
    assert np.allclose([float(a2d[2]), float(a2d[3])], pxo, rtol=0, atol=1e-6)     # printed with %.9g; the values are floats
    # the second Run of the driver used device-resident frames whose pyramids were built on the device
    # from level 0: the scene's pyramids are pyrDown chains, pyrDown is bit-exact -> the same pose, bit for bit
    assert int(out[4].split()[1]) == n
    Tres = np.array([float(v) for v in out[5].split()[1:]]).reshape(3, 4)
    assert np.array_equal(Tres, T)
    # the third Run re-read the map points before every level (one launch per level, as the reference's per-level
    # Get_Pose, src/Sprase_ImageAlign.cpp:84-103): same count and iterations, the pose to the rounding of the hand-over
    assert int(out[6].split()[1]) == n
    Tlv = np.array([float(v) for v in out[7].split()[1:]]).reshape(3, 4)
    assert [int(v) for v in out[8].split()[1:]] == iters
    assert np.abs(Tlv - T).max() < 1e-12
    # Feature_detector::detect of the C++ layer on the (feature-less) current frame vs the sequential restatement
    from tests import detector_restatement as R
    import math
    cell = 25
    cols, rows = math.ceil(320 / cell), math.ceil(240 / cell)
    cells = oracle.detect_cells(sc.cur_pyr, 3, cell, cols, rows, None, 5.0)
    want_det = R.detect(cells, 320, 240, cell, 200, [], [], 30)
    det = [int(v) for v in out[9].split()[1:]]
    assert det[0] == len(want_det) and det[0] > 10
    assert [tuple(det[1 + 3 * i:4 + 3 * i]) for i in range(det[0])] == want_det


def dump_world(path, cam, kfs, cur, mps, levels, cell, max_levels, frames=()):
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", levels, len(kfs), len(mps), cam.width, cam.height, cell, max_levels, len(frames)))
        f.write(struct.pack("<5f", cam.fx, cam.fy, cam.cx, cam.cy, cam.f))
        for fr in list(kfs) + [cur]:
            f.write(np.ascontiguousarray(fr.Get_Pose(), "<f8").tobytes())
            for l in fr.mvImg_Pyr[:levels]:
                f.write(np.ascontiguousarray(l).tobytes())
            f.write(struct.pack("<i", fr.n_features))
            for i in range(fr.n_features):
                f.write(fr.px[i].astype("<f4").tobytes() + struct.pack("<i", int(fr.level[i])) + fr.bearing[i].astype("<f8").tobytes())
        for mp in mps:
            f.write(np.asarray(mp.mPose, "<f8").tobytes() + struct.pack("<3i", mp.mnFound, int(mp.mbBad), len(mp.mObservations)))
            for k in sorted(mp.mObservations):
                f.write(struct.pack("<2i", k, mp.mObservations[k]))
        for img in frames:                                   # example_track: level-0 images of the frames to track
            f.write(np.ascontiguousarray(img, np.uint8).tobytes())


@pytest.mark.gpu
def test_cpp_search_local_points_matches_the_python_mirror(tmp_path, gpu_ctx):
    """Feature_Alignment::ResetGrid / ReprojectPoint / SearchLocalPoints of the C++ layer against
    dsdtm_amd.search.LocalPointSearch (itself held to the sequential restatement in test_search_gpu.py):
    same library calls underneath, so cells, map points, levels, pixels and the mask agree exactly."""
    from dsdtm_amd import search
    from dsdtm_amd.frame import Config
    from tests.test_search_gpu import make_world
    exe = build_example("example_search")
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5)
    cam, kfs, cur, mps = make_world(5, n_points=700)
    world = tmp_path / "world.bin"
    dump_world(world, cam, kfs, cur, mps, 5, 25, 5)
    out = subprocess.run([exe, str(world)], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    s = search.LocalPointSearch(cam, ctx=gpu_ctx)
    s.ResetGrid()
    n_in = sum(s.ReprojectPoint(cur, mp) for mp in mps)
    mask = np.full((cam.height, cam.width), 255, np.uint8)
    got = s.SearchLocalPoints(cur, kfs, mask)
    assert int(out[0].split()[1]) == n_in and int(out[1].split()[1]) == len(got) and len(got) > 100
    idx = {id(mp): i for i, mp in enumerate(mps)}
    for line, g in zip(out[2:2 + len(got)], got):
        c, mi, x, y, lv = line.split()
        assert (int(c), int(mi), int(lv)) == (g[0], idx[id(g[1])], g[3])
        assert np.float32(x) == g[2][0] and np.float32(y) == g[2][1]
    assert int(out[2 + len(got)].split()[1]) == int(mask.astype(np.uint64).sum())
    assert out[3 + len(got)].split() == ["resident_same", "1"]         # device-resident frames: one library call, same matches


@pytest.mark.gpu
def test_cpp_pose_optimization_matches_the_python_mirror(tmp_path, gpu_ctx):
    """Optimizer::PoseOptimization of the C++ layer against dsdtm_amd.Optimizer (itself held to the CPU
    restatement and a sequential restatement of the EraseFound walk in test_pose_opt_gpu.py): the same
    library call underneath, so pose, summary and every map-point counter agree exactly."""
    from dsdtm_amd import Optimizer
    from dsdtm_amd.frame import Frame
    from dsdtm_amd.search import MapPoint
    exe = build_example("example_pose_opt")
    cam = synth.Camera.tum()
    P = synth.make_pose_problem(91, n=260, cam=cam, outlier_frac=0.2, unused_frac=0.0, max_level=3)
    rng = np.random.default_rng(4)
    n = 260
    mps = [MapPoint(P.p_world[i].copy(), {}, int(rng.integers(1, 4)), bool(rng.random() < 0.05)) for i in range(n)]
    mp_idx = np.arange(n); mp_idx[rng.random(n) < 0.1] = -1
    initial = (rng.random(n) > 0.08).astype(np.uint8)
    with open(tmp_path / "frame.bin", "wb") as f:
        f.write(struct.pack("<2i", n, n) + struct.pack("<f", cam.f) + np.ascontiguousarray(P.T_seed, "<f8").tobytes())
        for mp in mps:
            f.write(np.asarray(mp.mPose, "<f8").tobytes() + struct.pack("<2i", mp.mnFound, int(mp.mbBad)))
        for i in range(n):
            f.write(P.bearing[i].astype("<f8").tobytes() + struct.pack("<3i", int(P.level[i]), int(initial[i]), int(mp_idx[i])))
    out = subprocess.run([exe, str(tmp_path / "frame.bin")], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    fr = Frame(cam, [np.zeros((8, 8), np.uint8)], P.T_seed)
    fr.set_features(np.zeros((n, 2), np.float32), P.bearing, P.p_world, initial, P.level)
    fr.mvMapPoints = [mps[k] if k >= 0 else None for k in mp_idx]
    sm = Optimizer.PoseOptimization(fr, 10, ctx=gpu_ctx)
    got = out[0].split()
    assert [int(v) for v in got[1:5]] == [sm["iterations"], sm["successful_steps"], sm["termination"], sm["n_residual_blocks"]]
    assert float(got[5]) == sm["initial_cost"] and float(got[6]) == sm["final_cost"] and sm["iterations"] > 3
    assert np.array_equal(np.array([float(v) for v in out[1].split()[1:]]), fr.Get_Pose().reshape(12))
    assert [tuple(int(v) for v in l.split()) for l in out[2:2 + n]] == [(mp.mnFound, int(mp.mbBad)) for mp in mps]
    assert sum(mp.mbBad for mp in mps) > 15                      # the walk erased something


@pytest.mark.gpu
def test_cpp_tracked_sequence_matches_the_python_mirror(tmp_path, gpu_ctx):
    """The tracked-frame loop (Run -> ResetGrid / ReprojectPoint / SearchLocalPoints -> PoseOptimization, every
    frame the reference of the next; src/Tracking.cpp:199-246) through the C++ classes of dsdtm_host.hpp against
    the same loop through the Python mirror (itself held to the oracle chain in test_tracking_sequence_gpu.py):
    six frames, same library underneath — tracked counts, match sets, refinement summaries and every map-point
    counter agree exactly, poses to rounding of the host-side projections."""
    import copy
    from dsdtm_amd import search
    from dsdtm_amd.frame import Config, Frame
    from dsdtm_amd.optimizer import Optimizer
    from dsdtm_amd.sparse_align import Sprase_ImgAlign
    from tests.test_search_gpu import make_world
    exe = build_example("example_track")
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    n_kf, n_frames = 2, 6
    cam, kfs, cur0, mps = make_world(33, n_points=600, n_kf=n_kf)
    tex = synth.make_texture(cam.height, cam.width, 33)
    rng = np.random.default_rng(5)
    T0 = np.vstack([kfs[-1].Get_Pose(), [0, 0, 0, 1]])
    imgs, xi = [], np.zeros(6)
    for k in range(n_frames):
        xi = xi + np.concatenate([rng.uniform(-0.012, 0.012, 3), rng.uniform(-0.006, 0.006, 3)])
        imgs.append(synth.warp_plane(tex, cam, synth.se3_exp(xi) @ T0, 2.0))
    world = tmp_path / "world.bin"
    dump_world(world, cam, kfs, cur0, mps, 5, 25, 5, frames=imgs)
    out = subprocess.run([exe, str(world)], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == 4 * n_frames
    # the same six frames with ONE library call per frame (Tracking::TrackFrame of the host layer -> dsdtm_track_frame: reprojection,
    # closest observation and the cell walk on the device instead of in the host layer): every printed line — poses to 17 digits,
    # matches, pixels, refinement summaries, map counters — is the same
    one = subprocess.run([exe, str(world), "onecall"], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert one == out
    # the Python mirror on the same world
    for k, kf in enumerate(kfs):
        mpts = [None] * kf.n_features
        for mp in mps:
            if k in mp.mObservations:
                mpts[mp.mObservations[k]] = mp
        kf.mvMapPoints = mpts
        kf.p_world = np.array([m.mPose if m is not None else np.zeros(3) for m in mpts])
        kf.initial = np.array([1 if m is not None else 0 for m in mpts], np.uint8)
    al = Sprase_ImgAlign(5, 0, 8, ctx=gpu_ctx, resident_frames=True)
    srch = search.LocalPointSearch(cam, ctx=gpu_ctx, resident_frames=True)
    idx = {id(mp): i for i, mp in enumerate(mps)}
    last = kfs[-1]
    for k in range(n_frames):
        cur = Frame(cam, synth.build_pyramid(imgs[k], 5), last.Get_Pose())
        n = al.Run(cur, last)
        t = out[4 * k].split()
        assert (int(t[1]), int(t[3])) == (k, n) and n >= 40
        assert np.allclose([float(v) for v in t[5:17]], cur.Get_Pose().reshape(12), rtol=0, atol=1e-9)
        srch.ResetGrid()
        for mp in mps:
            if not mp.IsBad():
                srch.ReprojectPoint(cur, mp)
        got = srch.SearchLocalPoints(cur, kfs)
        t = out[4 * k + 1].split()
        assert int(t[1]) == len(got) and len(got) >= 50
        cpp = [(int(t[2 + 5 * i]), int(t[3 + 5 * i]), int(t[4 + 5 * i])) for i in range(len(got))]
        assert cpp == [(g[0], idx[id(g[1])], g[3]) for g in got]
        cpx = np.array([[float(t[5 + 5 * i]), float(t[6 + 5 * i])] for i in range(len(got))])
        assert np.abs(cpx - np.array([g[2] for g in got])).max() < 1e-3
        sm = Optimizer.PoseOptimization(cur, ctx=gpu_ctx)
        t = out[4 * k + 2].split()
        assert (int(t[1]), int(t[2])) == (sm["iterations"], sm["termination"])
        assert np.allclose([float(v) for v in t[4:16]], cur.Get_Pose().reshape(12), rtol=0, atol=1e-9)
        assert out[4 * k + 3].split()[1:] == [f"{mp.mnFound}{'b' if mp.mbBad else ''}" for mp in mps]
        last = cur
