"""GPU parity at the FULL shape of every BASELINE.json config:
  config 1  an RGB-D sequence through the C++ driver in the shape of Test/test_SpraseImg_alignment.cpp
            (TUM data is not in the image: a generated RGB-D sequence stands in)
  config 2  tests/test_sparse_align_gpu.py::test_config2_pose_matches_oracle
  config 3  640x480, Tracking's (5, 0, 8), ~1000 patches, a stream of chained frames on device-resident frames
  config 4  the per-GPU share: ONE launch over 1024 independent 640x480 pairs, every pair against the oracle
  config 5  1280x960, 4 levels, 2000 patches: Run, then 2000 Align2D refinements on the same frame
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from dsdtm_amd import capi, synth
from tests import helpers as H
from tests.conftest import cached_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_share_1024_pairs_in_one_launch(gpu_ctx, oracle):
    """BASELINE config 4's per-GPU share (8192 pairs / 8 GPUs): 1024 independent 640x480 pairs, 4 levels,
    300 patches, cap 10, ONE launch of dsdtm_sparse_align_batch_device — every pose, tracked count,
    iteration count and exit code against the CPU oracle on the same bytes; then the same launch on a second
    stream while the first is still running (what bench.py --streams 2 does): bit-identical results."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    cam = synth.Camera.tum(640, 480)
    d = bench.build_batch(torch, dev, gpu_ctx, cam, 1024, 640, 480, 4, 300, seed=0xD5D7, stream=stream)
    cs = capi.camera_struct(cam)
    prm = capi.AlignParams(4, 0, 10, 15)
    d["T_cur_w"].copy_(d["T_seed"])
    torch.cuda.synchronize()
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stream.cuda_stream))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, stream.cuda_stream))
    Tg = d["T_cur_w"].cpu().numpy().copy()
    ntg = d["n_tracked"].cpu().numpy().copy()
    stg = np.frombuffer(d["stats"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE).copy()
    hb = bench.HostBatch(d, 1024)
    hb.run(oracle.load(), cs, prm, bench.usable_cpus())
    dl = np.array([synth.pose_error(Tg[i], hb.T[i]) for i in range(1024)])
    assert dl[:, 0].max() <= H.TIGHT_RAD and dl[:, 1].max() <= H.TIGHT_M, dl.max(axis=0)
    assert np.array_equal(ntg, hb.nt)
    for k in ("iters", "n_ref", "n_vis", "exit_code"):
        assert np.array_equal(stg[k], hb.st[k]), k
    assert np.allclose(stg["chi2"], hb.st["chi2"], rtol=1e-9, atol=0, equal_nan=True)
    err = np.array([synth.pose_error(Tg[i], d["T_true"][i]) for i in range(1024)])
    assert np.median(err[:, 0]) < 5e-4 and np.median(err[:, 1]) < 2e-3      # and they are alignments, not no-ops
    # two launches in flight on two streams (each stream has its own pair counters): same bits
    T2 = d["T_seed"].clone()
    d2 = capi.BatchDesc.from_buffer_copy(bytes(d["desc"]))
    d2.T_cur_w = T2.data_ptr()
    nt2 = torch.zeros_like(d["n_tracked"])
    d2.n_tracked, d2.stats = nt2.data_ptr(), None
    s2 = torch.cuda.Stream(device=dev)
    d["T_cur_w"].copy_(d["T_seed"])
    torch.cuda.synchronize()
    for rep in range(3):
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stream.cuda_stream))
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d2), C.byref(cs), C.byref(prm), s2.cuda_stream))
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, stream.cuda_stream))
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, s2.cuda_stream))
        assert np.array_equal(d["T_cur_w"].cpu().numpy(), Tg) and np.array_equal(T2.cpu().numpy(), Tg), rep
        assert np.array_equal(nt2.cpu().numpy(), ntg)
        d["T_cur_w"].copy_(d["T_seed"]); T2.copy_(d["T_seed"])
        torch.cuda.synchronize()


def test_config4_share_size_independent_properties(gpu_ctx):
    """The same 1024-pair launch held to properties that need no oracle (they hold for the reference's algorithm at any
    size): (1) GAUGE — moving the world frame (T_ref_w <- T_ref_w G, seeds likewise, map points <- G^-1 P) moves every
    result by exactly G (the path only ever uses T_cur T_ref^-1 and |P_w - C_ref|); (2) the ORDER of a pair's features
    does not matter beyond summation rounding; (3) RESTART — aligning again from the result comes back to it: the coarse
    levels have their own optima and walk away from it, the finest level returns to the same minimum (to the step size
    at which Gauss-Newton stops), and the error against the ground truth does not change."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    cam = synth.Camera.tum(640, 480)
    P, N = 1024, 300
    d = bench.build_batch(torch, dev, gpu_ctx, cam, P, 640, 480, 4, N, seed=0xD5D7, stream=stream)
    cs = capi.camera_struct(cam)
    prm = capi.AlignParams(4, 0, 10, 15)

    def run():
        torch.cuda.synchronize()
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stream.cuda_stream))
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, stream.cuda_stream))
        st = np.frombuffer(d["stats"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE).copy()
        return d["T_cur_w"].cpu().numpy().copy(), d["n_tracked"].cpu().numpy().copy(), st

    d["T_cur_w"].copy_(d["T_seed"])
    T0, n0, s0 = run()
    to44 = lambda T: np.concatenate([T.reshape(-1, 3, 4), np.tile(np.array([[[0.0, 0, 0, 1]]]), (len(T), 1, 1))], axis=1)
    # (1) gauge: one random rigid motion G per pair
    rng = np.random.default_rng(44)
    G = np.stack([np.vstack([synth.random_pose(rng, 2.0, 1.0), [0, 0, 0, 1]]) for _ in range(P)])
    Gi = np.linalg.inv(G)
    keep = {k: d[k].clone() for k in ("T_ref_w", "T_seed", "p_world")}
    Tref, Tseed = to44(keep["T_ref_w"].cpu().numpy()), to44(keep["T_seed"].cpu().numpy())
    pw = keep["p_world"].cpu().numpy()
    d["T_ref_w"].copy_(torch.from_numpy(np.ascontiguousarray((Tref @ G)[:, :3, :].reshape(P, 12))))
    d["T_cur_w"].copy_(torch.from_numpy(np.ascontiguousarray((Tseed @ G)[:, :3, :].reshape(P, 12))))
    d["p_world"].copy_(torch.from_numpy(np.ascontiguousarray(np.einsum("nij,npj->npi", Gi[:, :3, :3], pw) + Gi[:, None, :3, 3])))
    T1, n1, s1 = run()
    want = (to44(T0) @ G)[:, :3, :]
    dl = np.array([synth.pose_error(T1[i], want[i]) for i in range(P)])
    assert dl[:, 0].max() < 1e-9 and dl[:, 1].max() < 1e-8, dl.max(axis=0)
    assert np.array_equal(n1, n0) and (s1["iters"] == s0["iters"]).mean() > 0.99      # rounding may move a rare borderline decision
    for k, v in keep.items():
        d[k].copy_(v)
    # (2) feature order: reverse every pair's feature columns
    cols = {k: d[k].clone() for k in ("px", "bearing", "p_world", "initial")}
    for k, v in cols.items():
        d[k].copy_(torch.flip(v, dims=[1]))
    d["T_cur_w"].copy_(d["T_seed"])
    T2, n2, s2 = run()
    dl = np.array([synth.pose_error(T2[i], T0[i]) for i in range(P)])
    assert dl[:, 0].max() < 1e-9 and dl[:, 1].max() < 1e-8 and np.array_equal(n2, n0), dl.max(axis=0)
    for k, v in cols.items():
        d[k].copy_(v)
    # (3) restart from the result
    d["T_cur_w"].copy_(torch.from_numpy(T0))
    T3, n3, s3 = run()
    assert np.array_equal(n3, n0)
    dl = np.array([synth.pose_error(T3[i], T0[i]) for i in range(P)])
    # (measured: median 8e-8 rad / 2e-7 m, 99th percentile 2.4e-5 rad, maximum 9e-5 rad / 3e-4 m)
    assert np.median(dl[:, 0]) < 2e-6 and np.median(dl[:, 1]) < 5e-6 and np.percentile(dl[:, 0], 99) < 1e-4 and dl[:, 0].max() < 5e-4, \
        (np.median(dl, axis=0), dl.max(axis=0))
    e0 = np.array([synth.pose_error(T0[i], d["T_true"][i]) for i in range(P)])
    e3 = np.array([synth.pose_error(T3[i], d["T_true"][i]) for i in range(P)])
    assert abs(np.median(e3[:, 0]) - np.median(e0[:, 0])) < 2e-5 and abs(np.median(e3[:, 1]) - np.median(e0[:, 1])) < 5e-5


def test_config5_1280x960_2000_patches_then_align2d(gpu_ctx, oracle):
    """BASELINE config 5: 1280x960 pyramid, 4 levels, 2000 patches (one pair over a team of 8 compute units),
    then the per-feature Align2D refinement of all 2000 features on the same current frame (10x10 / 8x8
    patches cut from the reference at level 0, start offset U(-1.5, 1.5) px, 10 iterations)."""
    from dsdtm_amd import feature_alignment as FA
    sc = cached_scene(width=1280, height=960, levels=4, n_patches=2000, seed=0xC5, margin=40)
    To, no, so = oracle.sparse_align(sc, 4, 0, 10)
    for resident in (False, True):
        from dsdtm_amd.frame import Config, frames_from_scene
        from dsdtm_amd.sparse_align import Sprase_ImgAlign
        Config.Set("Camera.Min_fts", 15)
        cur, ref = frames_from_scene(sc)
        al = Sprase_ImgAlign(4, 0, 10, ctx=gpu_ctx, resident_frames=resident)
        ng = al.Run(cur, ref)
        H.assert_pose_close(cur.Get_Pose(), To, H.TIGHT_RAD, H.TIGHT_M, what=f"config 5 (resident={resident})")
        assert ng == no
        for k in ("iters", "n_ref", "n_vis", "exit_code"):
            assert al.last_stats[k] == so[k], (k, al.last_stats[k], so[k])
    ea, et = synth.pose_error(cur.Get_Pose(), sc.T_cur_w_true)
    assert ea < 2e-4 and et < 5e-4, (ea, et)
    # Align2D: every feature's reference patch at level 0, searched in the current level 0 around its true
    # position (the feature reprojected with the aligned pose) plus the start offset
    rng = np.random.default_rng(55)
    T = cur.Get_Pose()
    Pc = sc.p_world @ T[:, :3].T + T[:, 3]
    uv = np.stack([sc.cam.fx * Pc[:, 0] / Pc[:, 2] + sc.cam.cx, sc.cam.fy * Pc[:, 1] / Pc[:, 2] + sc.cam.cy], axis=1)
    keep = (uv[:, 0] > 12) & (uv[:, 0] < 1280 - 12) & (uv[:, 1] > 12) & (uv[:, 1] < 960 - 12)
    assert keep.sum() >= 1950
    pb, p = H.make_border_patches(sc.ref_pyr[0], [tuple(c) for c in sc.px[keep].astype(np.float64)])
    px0 = uv[keep] + rng.uniform(-1.5, 1.5, (keep.sum(), 2))
    level = np.zeros(keep.sum(), np.int32)
    co, pxo = oracle.align2d_batch(sc.cur_pyr, pb, p, level, px0, 10)
    cg, pxg = FA.align2d_batch(sc.cur_pyr, pb, p, level, px0, 10, ctx=gpu_ctx)
    assert np.array_equal(cg, co)
    assert np.array_equal(pxg[co], pxo[co])                      # exact-order float sums: identical pixels
    assert co.mean() > 0.9
    d = np.hypot(*(pxg[co] - uv[keep][co]).T)
    assert np.median(d) < 0.2, np.median(d)


def test_config3_hundred_chained_frames_of_1000_patches(gpu_ctx, oracle):
    """BASELINE config 3 as BASELINE.md §3 defines it: a 100-frame CHAINED 640x480 sequence — every frame is aligned
    against the previous one and its result seeds the next (src/Tracking.cpp:199-205: Run(mCurrentFrame, mLastFrame)
    with the current pose seeded from the last) — ~1000 patches per reference frame after the moving-object mask (a
    tenth of the features carry mbInitial = false, as masked-out features do), TUM fr3 intrinsics 535.4 / 539.2 /
    320.1 / 247.6 with Camera.f = 525 (Config/default.yaml:47-50,61), the live tracker's Sprase_ImgAlign(5, 0, 8)
    (src/Tracking.cpp:20-24,37), frames resident on the device (level 0 uploaded once, pyramid built there). The GPU
    chain and the oracle chain run side by side, each on its own poses; every frame is compared."""
    from dsdtm_amd.frame import Config, Frame
    from dsdtm_amd.sparse_align import Sprase_ImgAlign
    Config.Set("Camera.Min_fts", 15)
    W, Hh, L, N, K = 640, 480, 5, 1000, 100
    cam = synth.Camera(535.4, 539.2, 320.1, 247.6, 525.0, W, Hh)
    rng = np.random.default_rng(3003)
    tex = synth.make_texture(Hh, W, 3003)
    depth, nrm = 2.0, np.array([0.0, 0.0, 1.0])
    # a smooth path in front of the plane z = depth of the world (= frame 0): <= ~0.01 m / ~0.006 rad per frame
    amp = np.array([0.12, 0.10, 0.06, 0.05, 0.05, 0.05]); frq = rng.uniform(0.05, 0.09, 6); phs = rng.uniform(0, 2 * np.pi, 6)
    Tw = [synth.se3_exp(amp * (np.sin(frq * k + phs) - np.sin(phs))) for k in range(K + 1)]        # frame 0 = identity

    def features_of(k):                                          # features of frame k as a reference frame
        px = np.stack([rng.uniform(40, W - 40, N), rng.uniform(40, Hh - 40, N)], axis=1).astype(np.float32)
        bearing = synth.bearing_from_px(cam, px)
        R, t = Tw[k][:3, :3], Tw[k][:3, 3]
        Cw = -R.T @ t
        d = bearing @ R
        p_world = Cw + d * ((depth - nrm @ Cw) / (d @ nrm))[:, None]
        initial = (rng.random(N) >= 0.1).astype(np.uint8)
        return px, bearing, p_world, initial

    al = Sprase_ImgAlign(5, 0, 8, ctx=gpu_ctx, resident_frames=True)
    img0 = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    ref_g = Frame(cam, [img0], Tw[0][:3].copy())
    ref_g._device_frame = capi.DeviceFrame.from_image(gpu_ctx, img0, L)
    ref_pyr = synth.build_pyramid(img0, L)
    pose_g, pose_o = Tw[0][:3].copy(), Tw[0][:3].copy()          # the two chains' poses of the reference frame
    worst = (0.0, 0.0)
    for k in range(K):
        px, bearing, p_world, initial = features_of(k)
        img = synth.warp_plane(tex, cam, Tw[k + 1], depth)
        cur_pyr = synth.build_pyramid(img, L)
        # GPU chain: reference = the previous frame with ITS estimated pose, current seeded with that pose
        ref_g.Set_Pose(pose_g)
        ref_g.set_features(px, bearing, p_world, initial)
        cur_g = Frame(cam, [img], pose_g.copy())
        cur_g._device_frame = capi.DeviceFrame.from_image(gpu_ctx, img, L)
        ng = al.Run(cur_g, ref_g)
        # oracle chain
        sc = synth.AlignScene(cam, ref_pyr, cur_pyr, px, bearing, p_world, initial, pose_o.copy(), pose_o.copy(), Tw[k + 1][:3], depth)
        To, no, so = oracle.sparse_align(sc, 5, 0, 8)
        H.assert_pose_close(cur_g.Get_Pose(), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"frame {k + 1}")
        assert ng == no and al.last_stats["iters"] == so["iters"] and al.last_stats["exit_code"] == so["exit_code"], k
        assert al.last_stats["n_ref"][0] == so["n_ref"][0] and 850 < no <= N
        ea, et = synth.pose_error(cur_g.Get_Pose(), Tw[k + 1][:3])
        worst = (max(worst[0], ea), max(worst[1], et))
        ref_g._device_frame.close()
        ref_g, ref_pyr = cur_g, cur_pyr
        pose_g, pose_o = cur_g.Get_Pose().copy(), To.copy()
    ref_g._device_frame.close()
    assert worst[0] < 2e-3 and worst[1] < 5e-3, worst            # the chain does not drift away from the ground truth


def test_config1_rgbd_sequence_through_the_cpp_driver(gpu_ctx, oracle, tmp_path):
    """BASELINE config 1 (TUM fr1/xyz through Test/test_SpraseImg_alignment.cpp) with a generated RGB-D
    sequence standing in for the dataset: dsdtm_amd/host/example_rgbd.cpp follows the reference test line
    by line — detector on the first frame, 3-D points from the depth map, Sprase_ImgAlign(4, 0, 30),
    Run(cur, ref) per frame seeded with the previous pose, error against ground truth printed — and the
    poses it prints are held to the CPU oracle run on the very features the driver used."""
    from tests.test_host_cpp import build_example
    exe = build_example("example_rgbd")
    W, Hh, L, depth, n_frames = 640, 480, 4, 2.0, 5
    cam = synth.Camera.tum(W, Hh)
    tex = synth.make_texture(Hh, W, 0x7A)
    rng = np.random.default_rng(11)
    T_ref = synth.random_pose(rng)
    T4 = np.vstack([T_ref, [0, 0, 0, 1]])
    frames, poses = [np.clip(np.rint(tex), 0, 255).astype(np.uint8)], [T_ref]
    xi = np.zeros(6)
    for k in range(1, n_frames):
        xi = xi + np.concatenate([rng.uniform(-0.01, 0.01, 3), rng.uniform(-0.005, 0.005, 3)])
        T_cr = synth.se3_exp(xi)
        frames.append(synth.warp_plane(tex, cam, T_cr, depth))
        poses.append((T_cr @ T4)[:3])
    seq = tmp_path / "seq.bin"
    with open(seq, "wb") as f:
        f.write(struct.pack("<6i", W, Hh, L, n_frames, 20, 400))
        f.write(struct.pack("<5f", cam.fx, cam.fy, cam.cx, cam.cy, cam.f))
        for k in range(n_frames):
            f.write(frames[k].tobytes() + np.ascontiguousarray(poses[k], "<f8").tobytes())
            if k == 0:
                f.write(np.full((Hh, W), depth, "<f4").tobytes())     # blender-style z-depth of the plane
    feats = tmp_path / "features.bin"
    out = subprocess.run([exe, str(seq), str(feats)], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    n = int(out[0].split()[1])
    assert n >= 100, out[0]
    raw = np.fromfile(feats, dtype=np.uint8)[4:].reshape(n, 56)
    px = raw[:, :8].copy().view("<f4").reshape(n, 2)
    bearing = raw[:, 8:32].copy().view("<f8").reshape(n, 3)
    p_world = raw[:, 32:56].copy().view("<f8").reshape(n, 3)
    # the 3-D points the driver made from the depth map lie on the plane z = depth of the reference camera
    Pc = p_world @ T_ref[:, :3].T + T_ref[:, 3]
    assert np.abs(Pc[:, 2] - depth).max() < 1e-5
    sc = type("S", (), {})()
    sc.cam, sc.ref_pyr = cam, synth.build_pyramid(frames[0], L)
    sc.px, sc.bearing, sc.p_world, sc.initial = px, bearing, p_world, np.ones(n, np.uint8)
    sc.T_ref_w = T_ref
    To = T_ref.copy()
    for k in range(1, n_frames):
        tok = out[k].split()
        assert tok[0] == "frame" and int(tok[1]) == k
        tracked, te, ang = int(tok[3]), float(tok[5]), float(tok[7])
        T = np.array([float(v) for v in tok[9:21]]).reshape(3, 4)
        iters = [int(v) for v in tok[22:26]]
        sc.cur_pyr = synth.build_pyramid(frames[k], L)
        sc.T_cur_w_seed = To
        To, no, so = oracle.sparse_align(sc, 4, 0, 30)
        H.assert_pose_close(T, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"rgbd frame {k}")
        assert tracked == no and iters == so["iters"][:4]
        assert te < 2e-3 and ang < 5e-4, (k, te, ang)                  # the error against ground truth the test prints


def test_config4_whole_8192_pairs_over_eight_contexts(gpu_ctx, oracle):
    """BASELINE config 4 WHOLE, on one GPU: 8192 independent 640x480 pairs — the eight 1024-pair blocks bench.py's ranks
    0..7 build (seeds shard.batch_seed(0xD5D7, r)), 4 levels, 300 patches, cap 10 — as ONE host batch through
    dsdtm_sparse_align_batch_sharded with EIGHT contexts on device 0 (one host thread + stream per context: the one-process
    form of the 8-GPU split, src/Sprase_ImageAlign.cpp:29-60 per pair) and, level 0 only, through
    dsdtm_sparse_align_batch_streamed. Every pose / n_tracked / iteration count / exit code against the CPU oracle on the
    same bytes (all host threads), both entries bit-equal to each other and to eight single-context device launches."""
    import torch
    import bench
    from dsdtm_amd import shard
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    W, Hh, L, N, PB, G = 640, 480, 4, 300, 1024, 8
    P = PB * G
    cam = synth.Camera.tum(W, Hh)
    cs = capi.camera_struct(cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    lib = gpu_ctx.lib
    ws, hs, st_, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    pitch = (nbytes + 255) // 256 * 256
    host = dict(ref=np.empty((P, pitch), np.uint8), cur=np.empty((P, pitch), np.uint8), px=np.empty((P, N, 2), np.float32),
                bear=np.empty((P, N, 3)), pw=np.empty((P, N, 3)), ini=np.empty((P, N), np.uint8), Tr=np.empty((P, 12)),
                seed=np.empty((P, 12)))
    single = dict(T=np.empty((P, 12)), nt=np.empty(P, np.int32), st=np.empty(P, capi.STATS_DTYPE))
    for r in range(G):
        d = bench.build_batch(torch, dev, gpu_ctx, cam, PB, W, Hh, L, N, seed=shard.batch_seed(0xD5D7, r), stream=stream)
        assert d["pitch"] == pitch
        lo, hi = shard.pair_range(P, r, G)
        assert (lo, hi) == (r * PB, (r + 1) * PB)
        for k, src in (("ref", "ref_pyr"), ("cur", "cur_pyr"), ("px", "px"), ("bear", "bearing"), ("pw", "p_world"), ("ini", "initial"),
                       ("Tr", "T_ref_w"), ("seed", "T_seed")):
            host[k][lo:hi] = d[src].cpu().numpy()
        # the block as ONE single-context device launch (what rank r of an 8-GPU run does)
        d["T_cur_w"].copy_(d["T_seed"])
        torch.cuda.synchronize()
        gpu_ctx.check(lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stream.cuda_stream))
        gpu_ctx.check(lib.dsdtm_sparse_align_check(gpu_ctx.handle, stream.cuda_stream))
        single["T"][lo:hi] = d["T_cur_w"].cpu().numpy()
        single["nt"][lo:hi] = d["n_tracked"].cpu().numpy()
        single["st"][lo:hi] = np.frombuffer(d["stats"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
        del d
        torch.cuda.empty_cache()

    def desc_for(T, nt, st):
        b = capi.BatchDesc()
        b.n_pairs, b.max_features, b.levels = P, N, L
        for l in range(L):
            b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], st_[l], offs[l]
        b.pyr_pitch = pitch
        b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (host[k].ctypes.data for k in ("ref", "cur", "px", "bear", "pw"))
        b.initial, b.n_features, b.T_ref_w, b.T_cur_w = host["ini"].ctypes.data, None, host["Tr"].ctypes.data, T.ctypes.data
        b.n_tracked, b.stats = nt.ctypes.data, st.ctypes.data
        return b

    # the CPU oracle on the same bytes, every host thread (8192 alignments: ~0.5 s on 16 threads)
    To, nto, sto = host["seed"].copy(), np.zeros(P, np.int32), np.zeros(P, capi.STATS_DTYPE)
    bo = desc_for(To, nto, sto)
    oracle.load().oracle_sparse_align_batch_timed(C.byref(bo), C.byref(cs), C.byref(prm), bench.usable_cpus())

    ctxs = [capi.Context(0) for _ in range(G)]
    arr = (C.c_void_p * G)(*[c.handle for c in ctxs])
    # 1. whole pyramids from host memory, eight shards
    Ts, nts, sts = host["seed"].copy(), np.full(P, -1, np.int32), np.zeros(P, capi.STATS_DTYPE)
    bs = desc_for(Ts, nts, sts)
    rc = lib.dsdtm_sparse_align_batch_sharded(arr, G, C.byref(bs), C.byref(cs), C.byref(prm))
    assert rc == 0, [c.lib.dsdtm_last_error(c.handle) for c in ctxs]
    # 2. level 0 only (the pyramids' first W*H bytes: image_pitch = the pyramid pitch), pyramids on the device, chunks of 128
    Tt, ntt, stt = host["seed"].copy(), np.full(P, -1, np.int32), np.zeros(P, capi.STATS_DTYPE)
    s = capi.StreamDesc()
    s.n_pairs, s.max_features, s.levels, s.width, s.height, s.row_stride, s.image_pitch = P, N, L, W, Hh, W, pitch
    s.ref_image, s.cur_image = host["ref"].ctypes.data, host["cur"].ctypes.data
    s.px_xy, s.bearing, s.p_world, s.initial = (host[k].ctypes.data for k in ("px", "bear", "pw", "ini"))
    s.n_features, s.T_ref_w, s.T_cur_w, s.n_tracked, s.stats = None, host["Tr"].ctypes.data, Tt.ctypes.data, ntt.ctypes.data, stt.ctypes.data
    rc = lib.dsdtm_sparse_align_batch_streamed(arr, G, C.byref(s), 128, C.byref(cs), C.byref(prm))
    assert rc == 0, [c.lib.dsdtm_last_error(c.handle) for c in ctxs]
    for c in ctxs:
        c.close()

    # both host-fed entries and the eight single-context launches: the same bits
    for what, (T, nt, st) in (("sharded", (Ts, nts, sts)), ("streamed", (Tt, ntt, stt))):
        assert np.array_equal(T, single["T"]), what
        assert np.array_equal(nt, single["nt"]), what
        for k in ("iters", "n_ref", "n_vis", "exit_code", "chi2"):
            assert np.array_equal(st[k], single["st"][k], equal_nan=(k == "chi2")), (what, k)
    # and the oracle's results, pair by pair
    dl = np.array([synth.pose_error(Ts[i].reshape(3, 4), To[i].reshape(3, 4)) for i in range(P)])
    assert np.isfinite(dl).all() and dl[:, 0].max() <= H.TIGHT_RAD and dl[:, 1].max() <= H.TIGHT_M, dl.max(axis=0)
    assert np.array_equal(nts, nto)
    for k in ("iters", "n_ref", "n_vis", "exit_code"):
        assert np.array_equal(sts[k], sto[k]), k
    assert np.allclose(sts["chi2"], sto["chi2"], rtol=1e-9, atol=0, equal_nan=True)
    assert (nts > 250).mean() > 0.99                      # alignments, not no-ops
    # the eight blocks differ (a rank that re-ran block 0 would not go unnoticed)
    assert len({host["Tr"][r * PB].tobytes() for r in range(G)}) == G
