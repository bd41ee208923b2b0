"""dsdtm_track_frame against the four-call chain over many tracked sequences: worlds of different sizes, seeds and cell sizes,
every frame's Run pose / count / iterations, match list, refined pose and map side effects compared bit for bit.
usage: python tools/soak_track.py [n_sequences]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsdtm_amd import capi, search, synth, tracking
from dsdtm_amd.frame import Config, Frame
from dsdtm_amd.optimizer import Optimizer
from dsdtm_amd.sparse_align import Sprase_ImgAlign
from tests.test_search_gpu import make_world

torch.cuda.init()
ctx = capi.default_context(0)
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng0 = np.random.default_rng(2026)
bad = frames = scans = 0
t0 = time.time()
for q in range(n_seq):
    cell = int(rng0.choice([15, 20, 25, 32]))
    n_points = int(rng0.choice([300, 700, 1200, 2500]))
    n_kf = int(rng0.choice([2, 3]))
    Config.Set("Camera.CellSize", cell); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    seed = 500 + q
    cam, kfs, _, mps = make_world(seed, n_points=n_points, n_kf=n_kf, cell=cell)
    tex = synth.make_texture(cam.height, cam.width, seed)
    for k, kf in enumerate(kfs):
        mpts = [None] * kf.n_features
        for mp in mps:
            if k in mp.mObservations:
                mpts[mp.mObservations[k]] = mp
        kf.mvMapPoints = mpts
        kf.p_world = np.array([m.mPose if m is not None else np.zeros(3) for m in mpts])
        kf.initial = np.array([1 if m is not None else 0 for m in mpts], np.uint8)
    worlds = [copy.deepcopy((kfs, mps)) for _ in range(2)]
    rng = np.random.default_rng(seed)
    T0 = np.vstack([kfs[-1].Get_Pose(), [0, 0, 0, 1]])
    imgs, xi = [], np.zeros(6)
    for k in range(5):
        xi = xi + np.concatenate([rng.uniform(-0.012, 0.012, 3), rng.uniform(-0.006, 0.006, 3)])
        imgs.append(synth.warp_plane(tex, cam, synth.se3_exp(xi) @ T0, 2.0))
    a_kfs, a_mps = worlds[0]
    al = Sprase_ImgAlign(5, 0, 8, ctx=ctx, resident_frames=True)
    srch = search.LocalPointSearch(cam, ctx=ctx, resident_frames=True)
    a_idx = {id(mp): i for i, mp in enumerate(a_mps)}
    log, last = [], a_kfs[-1]
    for k in range(5):
        cur = Frame(cam, synth.build_pyramid(imgs[k], 5), last.Get_Pose())
        n = al.Run(cur, last)
        T_run = cur.Get_Pose().copy()
        srch.ResetGrid()
        for mp in a_mps:
            if not mp.IsBad():
                srch.ReprojectPoint(cur, mp)
        ms = srch.SearchLocalPoints(cur, a_kfs) if n >= 20 else []
        if n >= 20:
            Optimizer.PoseOptimization(cur, ctx=ctx)
        log.append((n, T_run, list(al.last_stats["iters"]), [(m[0], a_idx[id(m[1])], float(m[2][0]), float(m[2][1]), m[3]) for m in ms],
                    cur.Get_Pose().copy(), [mp.mnFound for mp in a_mps], [mp.mbBad for mp in a_mps]))
        last = cur
    b_kfs, b_mps = worlds[1]
    trk = tracking.Tracker(cam, ctx=ctx, max_level=5, min_level=0, max_iters=8, min_tracked=20)
    b_idx = {id(mp): i for i, mp in enumerate(b_mps)}
    last = b_kfs[-1]
    for k in range(5):
        cur, n, ms = trk.TrackFrame(imgs[k], last, b_kfs, b_mps)
        r, a = trk.last_result, log[k]
        got = (n, r["T_run"], list(r["stats"]["iters"]), [(m[0], b_idx[id(m[1])], float(m[2][0]), float(m[2][1]), m[3]) for m in ms],
               cur.Get_Pose(), [mp.mnFound for mp in b_mps], [mp.mbBad for mp in b_mps])
        same = (got[0] == a[0] and np.array_equal(got[1], a[1]) and got[2] == a[2] and got[3] == a[3] and np.array_equal(got[4], a[4])
                and got[5] == a[5] and got[6] == a[6])
        frames += 1
        scans += int(r["replay_full_scan"])
        if not same:
            bad += 1
            print(f"sequence {q} (cell {cell}, {n_points} points, {n_kf} keyframes) frame {k}: DIFFERENT", flush=True)
        last = cur
    print(f"sequence {q:2d}: cell {cell:2d}, {n_points:4d} points, {n_kf} keyframes: tracked {[l[0] for l in log]}, matches {[len(l[3]) for l in log]}", flush=True)
print(f"{n_seq} sequences, {frames} frames in {time.time() - t0:.1f} s: {bad} frames differ from the four-call chain; the replay ran its full scan on {scans} frames")
sys.exit(1 if bad else 0)
