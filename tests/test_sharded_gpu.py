"""Independent frame pairs over several contexts / devices behind the C ABI (dsdtm_sparse_align_batch_sharded,
SURVEY.md §8(b)/(e): contiguous blocks of ceil(P/G) pairs, one host thread and stream per context, no collective),
and the chained batch (frame k is `cur` of pair k - 1 and `ref` of pair k: cur_pyr == ref_pyr + pyr_pitch)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from dsdtm_amd import capi, shard, synth
from tests import helpers as H
from tests.conftest import cached_scene
from tests.test_host_cpp import build_example


def test_shard_range_is_contiguous_ceil_blocks():
    """The C entry and the Python helper the bench uses cut the same blocks; together they cover every pair once."""
    lib = capi.load()
    for P in (0, 1, 7, 8, 9, 1000, 1024, 8192):
        for G in (1, 2, 3, 8, 16):
            seen = []
            for g in range(G):
                lo, hi = C.c_int(), C.c_int()
                lib.dsdtm_shard_range(P, G, g, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard.pair_range(P, g, G)
                assert hi.value - lo.value <= -(-P // G)
                seen += list(range(lo.value, hi.value))
            assert seen == list(range(P))
    lo, hi = C.c_int(5), C.c_int(5)
    lib.dsdtm_shard_range(10, 4, 7, C.byref(lo), C.byref(hi))           # a shard index out of range is empty
    assert lo.value == hi.value


def _host_batch(scenes, L, W, Hh):
    ws, hs, st, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    pitch = (nbytes + 255) // 256 * 256
    P, N = len(scenes), len(scenes[0].px)
    a = dict(ref=np.zeros((P, pitch), np.uint8), cur=np.zeros((P, pitch), np.uint8))
    for i, sc in enumerate(scenes):
        for l in range(L):
            a["ref"][i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.ref_pyr[l].reshape(-1)
            a["cur"][i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.cur_pyr[l].reshape(-1)
    a.update(px=np.ascontiguousarray(np.stack([s.px for s in scenes]), np.float32),
             bear=np.ascontiguousarray(np.stack([s.bearing for s in scenes])),
             pw=np.ascontiguousarray(np.stack([s.p_world for s in scenes])),
             ini=np.ascontiguousarray(np.stack([s.initial for s in scenes]), np.uint8),
             Tr=np.ascontiguousarray(np.stack([s.T_ref_w.reshape(12) for s in scenes])),
             Tc=np.ascontiguousarray(np.stack([s.T_cur_w_seed.reshape(12) for s in scenes])),
             nt=np.full(P, -1, np.int32), st=np.zeros(P, capi.STATS_DTYPE))
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = P, N, L
    for l in range(L):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], st[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (a[k].ctypes.data for k in ("ref", "cur", "px", "bear", "pw"))
    b.initial, b.n_features, b.T_ref_w, b.T_cur_w = a["ini"].ctypes.data, None, a["Tr"].ctypes.data, a["Tc"].ctypes.data
    b.n_tracked, b.stats = a["nt"].ctypes.data, a["st"].ctypes.data
    return a, b, pitch


@pytest.mark.gpu
def test_two_and_three_contexts_on_one_device_equal_one_context(gpu_ctx, oracle):
    """The one-GPU rehearsal of the 8-GPU split: the batch over 1, 2 and 3 contexts (all on device 0, each with its
    own host thread and stream) — identical results bit for bit, and the oracle's."""
    W, Hh, L = 320, 240, 3
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=150, seed=4100 + i, margin=12) for i in range(11)]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    lib = gpu_ctx.lib
    results = []
    for G in (1, 2, 3):
        ctxs = [capi.Context(0) for _ in range(G)]
        a, b, _ = _host_batch(scenes, L, W, Hh)
        arr = (C.c_void_p * G)(*[c.handle for c in ctxs])
        rc = lib.dsdtm_sparse_align_batch_sharded(arr, G, C.byref(b), C.byref(cam), C.byref(prm))
        assert rc == 0, ctxs[0].lib.dsdtm_last_error(ctxs[0].handle)
        results.append((a["Tc"].copy(), a["nt"].copy(), a["st"].copy()))
        for c in ctxs:
            c.close()
    for T, nt, st in results[1:]:
        assert np.array_equal(T, results[0][0]) and np.array_equal(nt, results[0][1])
        assert np.array_equal(st["iters"], results[0][2]["iters"]) and np.array_equal(st["chi2"], results[0][2]["chi2"])
    for i, sc in enumerate(scenes):
        To, no, so = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(results[0][0][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")
        assert results[0][1][i] == no and list(results[0][2]["iters"][i][:L]) == list(so["iters"][:L])
    # a context listed twice is refused (a context is single-threaded)
    c0 = capi.Context(0)
    a, b, _ = _host_batch(scenes, L, W, Hh)
    arr = (C.c_void_p * 2)(c0.handle, c0.handle)
    assert lib.dsdtm_sparse_align_batch_sharded(arr, 2, C.byref(b), C.byref(cam), C.byref(prm)) != 0
    c0.close()


@pytest.mark.gpu
def test_example_batch_cpp_over_several_contexts(tmp_path, gpu_ctx, oracle):
    """The C++ caller (dsdtm_amd/host/example_batch.cpp: one process, one context per shard) against the oracle."""
    exe = build_example("example_batch")
    W, Hh, L = 320, 240, 3
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=150, seed=4100 + i, margin=12) for i in range(7)]
    a, b, pitch = _host_batch(scenes, L, W, Hh)
    path = tmp_path / "batch.bin"
    cam = scenes[0].cam
    with open(path, "wb") as f:
        f.write(struct.pack("<10i", len(scenes), 150, L, W, Hh, L, 0, 10, 15, pitch))
        f.write(struct.pack("<5f", cam.fx, cam.fy, cam.cx, cam.cy, cam.f))
        for k in ("ref", "cur", "px", "bear", "pw", "ini", "Tr", "Tc"):
            f.write(a[k].tobytes())
    out = subprocess.run([exe, str(path), "3"], capture_output=True, text=True, check=True).stdout.split("\n")
    # (default: the streamed entry — level 0 only, pyramids on the device; "sharded": whole pyramids. Same bits.)
    assert subprocess.run([exe, str(path), "3", "sharded"], capture_output=True, text=True, check=True).stdout.split("\n") == out
    shards = [l.split() for l in out if l.startswith("shard")]
    assert [(int(s[5]), int(s[6])) for s in shards] == [shard.pair_range(7, g, 3) for g in range(3)]
    rows = [l.split() for l in out if l.startswith("pair")]
    assert len(rows) == len(scenes)
    for i, sc in enumerate(scenes):
        To, no, so = oracle.sparse_align(sc, L, 0, 10)
        r = rows[i]
        assert int(r[3]) == no and [int(x) for x in r[5:5 + L]] == list(so["iters"][:L])
        T = np.array([float(x) for x in r[6 + L:6 + L + 12]]).reshape(3, 4)
        H.assert_pose_close(T, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")


@pytest.mark.gpu
def test_chained_batch_equals_per_pair_runs(gpu_ctx, oracle):
    """A sequence of frames as ONE device array of pyramids: ref_pyr = frame 0, cur_pyr = ref_pyr + pyr_pitch, so frame
    k is `cur` of pair k - 1 and `ref` of pair k and every frame is uploaded (and its pyramid built) once. Bit for bit
    the results of the same pairs with separate copies of their frames, and the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    W, Hh, L, N, K = 320, 240, 3, 150, 10
    seq = synth.make_sequence(n_frames=K, width=W, height=Hh, levels=L, n_patches=N, seed=21, margin=12)
    a, b, pitch = _host_batch(seq, L, W, Hh)
    cam = capi.camera_struct(seq[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    P = K - 1
    # unchained reference run: separate ref / cur arrays
    t = {k: torch.from_numpy(a[k]).to(dev) for k in ("ref", "cur", "px", "bear", "pw", "ini", "Tr", "Tc")}
    t["nt"] = torch.zeros(P, dtype=torch.int32, device=dev)
    t["st"] = torch.zeros((P, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    d = capi.BatchDesc.from_buffer_copy(bytes(b))
    d.ref_pyr, d.cur_pyr, d.px_xy, d.bearing, d.p_world = (t[k].data_ptr() for k in ("ref", "cur", "px", "bear", "pw"))
    d.initial, d.T_ref_w, d.T_cur_w, d.n_tracked, d.stats = (t[k].data_ptr() for k in ("ini", "Tr", "Tc", "nt", "st"))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d), C.byref(cam), C.byref(prm), None))
    torch.cuda.synchronize()
    want = (t["Tc"].cpu().numpy().copy(), t["nt"].cpu().numpy().copy(), t["st"].cpu().numpy().copy())
    # chained: K frames, level 0 uploaded, pyramids built on the device by the library's pyrDown
    ws, hs, st_, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    frames = np.zeros((K, pitch), np.uint8)
    for k in range(K):
        img = (seq[k].ref_pyr if k < P else seq[P - 1].cur_pyr)[0]
        frames[k, :W * Hh] = img.reshape(-1)
    fr = torch.from_numpy(frames).to(dev)
    wa, ha, sa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*st_)
    oa = (C.c_size_t * L)(*offs)
    gpu_ctx.check(gpu_ctx.lib.dsdtm_pyrdown_batch_device(gpu_ctx.handle, fr.data_ptr(), pitch, K, L, wa, ha, sa, oa, None))
    t["Tc"].copy_(torch.from_numpy(a["Tc"]).to(dev)); t["nt"].zero_(); t["st"].zero_()
    d.ref_pyr, d.cur_pyr = fr.data_ptr(), fr.data_ptr() + pitch
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(d), C.byref(cam), C.byref(prm), None))
    torch.cuda.synchronize()
    got = (t["Tc"].cpu().numpy(), t["nt"].cpu().numpy(), t["st"].cpu().numpy())
    assert np.array_equal(fr.cpu().numpy()[:P], a["ref"])              # the device pyramids are the host ones, byte for byte
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    for i, sc in enumerate(seq):
        To, no, _ = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(got[0][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"chained pair {i}")
        assert got[1][i] == no


@pytest.mark.gpu
def test_more_contexts_than_pairs_and_ragged_feature_counts(gpu_ctx, oracle):
    """Shards may be empty (5 contexts, 3 pairs) and pairs may carry different live feature counts (n_features)."""
    W, Hh, L = 320, 240, 3
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=150, seed=4100 + i, margin=12) for i in range(3)]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    a, b, _ = _host_batch(scenes, L, W, Hh)
    nf = np.array([150, 90, 10], np.int32)                       # the last one is below Min_fts: n_tracked 0, pose untouched
    b.n_features = nf.ctypes.data
    ctxs = [capi.Context(0) for _ in range(5)]
    arr = (C.c_void_p * 5)(*[c.handle for c in ctxs])
    assert gpu_ctx.lib.dsdtm_sparse_align_batch_sharded(arr, 5, C.byref(b), C.byref(cam), C.byref(prm)) == 0
    for c in ctxs:
        c.close()
    import copy
    for i, sc in enumerate(scenes):
        s2 = copy.copy(sc)
        n = int(nf[i])
        s2.px, s2.bearing, s2.p_world, s2.initial = sc.px[:n], sc.bearing[:n], sc.p_world[:n], sc.initial[:n]
        To, no, _ = oracle.sparse_align(s2, L, 0, 10)
        assert a["nt"][i] == no
        H.assert_pose_close(a["Tc"][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")
    assert a["nt"][2] == 0 and np.array_equal(a["Tc"][2], scenes[2].T_cur_w_seed.reshape(12))


@pytest.mark.gpu
def test_twenty_streams_through_one_context(gpu_ctx, oracle):
    """A context keeps launch bookkeeping (pair counters, scratch) for 16 streams; a 17th takes over the entry that has
    been idle longest once the event behind that entry's last launch has fired. 20 streams in turn, three rounds:
    every launch gives the oracle's results."""
    import torch
    from tests.test_sparse_align_gpu import _device_batch
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=150, seed=4100 + i, margin=12) for i in range(4)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in scenes]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    ctx = capi.Context(0)                                          # its own context: the table starts empty
    streams = [torch.cuda.Stream(device=dev) for _ in range(20)]
    packed = [_device_batch(torch, dev, scenes, L, W, Hh) for _ in streams]
    seed = torch.from_numpy(np.stack([s.T_cur_w_seed.reshape(12) for s in scenes])).to(dev)
    for rnd in range(3):
        for (t, b), st in zip(packed, streams):
            with torch.cuda.stream(st):
                t["Tc"].copy_(seed, non_blocking=True)
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
        torch.cuda.synchronize()
        for k, (t, b) in enumerate(packed):
            Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
            for i, (To, no, _) in enumerate(want):
                H.assert_pose_close(Tg[i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"round {rnd} stream {k} pair {i}")
                assert ntg[i] == no
    ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, streams[0].cuda_stream))
    ctx.close()


def _stream_desc(a, frames_ref, frames_cur, P, N, L, W, Hh, row_stride=None, image_pitch=None):
    s = capi.StreamDesc()
    s.n_pairs, s.max_features, s.levels, s.width, s.height = P, N, L, W, Hh
    s.row_stride = row_stride or W
    s.image_pitch = image_pitch or W * Hh
    s.ref_image = frames_ref.ctypes.data
    s.cur_image = frames_cur.ctypes.data if frames_cur is not None else None
    s.px_xy, s.bearing, s.p_world, s.initial = (a[k].ctypes.data for k in ("px", "bear", "pw", "ini"))
    s.n_features, s.T_ref_w, s.T_cur_w, s.n_tracked, s.stats = None, a["Tr"].ctypes.data, a["Tc"].ctypes.data, a["nt"].ctypes.data, a["st"].ctypes.data
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("chained", [False, True], ids=["separate ref/cur frames", "chained sequence"])
def test_streamed_batch_equals_the_sharded_entry(gpu_ctx, oracle, chained):
    """dsdtm_sparse_align_batch_streamed: level 0 only from host memory, chunks of 4 pairs (upload of chunk j + 1 beside
    pyramids + alignment of chunk j), pyramids by the device pyrDown — over 1, 2 and 3 contexts on one device. Bit for bit
    the results of dsdtm_sparse_align_batch_sharded on the same frames with host-built pyramids, i.e. the oracle's."""
    W, Hh, L, N = 320, 240, 3, 150
    if chained:
        scenes = synth.make_sequence(n_frames=12, width=W, height=Hh, levels=L, n_patches=N, seed=23, margin=12)
    else:
        scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=4100 + i, margin=12) for i in range(11)]
    P = len(scenes)
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    lib = gpu_ctx.lib
    a0, b0, _ = _host_batch(scenes, L, W, Hh)
    one = (C.c_void_p * 1)(gpu_ctx.handle)
    assert lib.dsdtm_sparse_align_batch_sharded(one, 1, C.byref(b0), C.byref(cam), C.byref(prm)) == 0
    if chained:
        fr = np.stack([sc.ref_pyr[0] for sc in scenes] + [scenes[-1].cur_pyr[0]]).reshape(P + 1, W * Hh)
        fc = None
    else:
        fr = np.stack([sc.ref_pyr[0] for sc in scenes]).reshape(P, W * Hh)
        fc = np.stack([sc.cur_pyr[0] for sc in scenes]).reshape(P, W * Hh)
    for G in (1, 2, 3):
        ctxs = [capi.Context(0) for _ in range(G)]
        a, _, _ = _host_batch(scenes, L, W, Hh)
        s = _stream_desc(a, fr, fc, P, N, L, W, Hh)
        arr = (C.c_void_p * G)(*[c.handle for c in ctxs])
        rc = lib.dsdtm_sparse_align_batch_streamed(arr, G, C.byref(s), 4, C.byref(cam), C.byref(prm))
        assert rc == 0, ctxs[0].lib.dsdtm_last_error(ctxs[0].handle)
        assert np.array_equal(a["Tc"], a0["Tc"]) and np.array_equal(a["nt"], a0["nt"])
        assert np.array_equal(a["st"]["iters"], a0["st"]["iters"]) and np.array_equal(a["st"]["chi2"], a0["st"]["chi2"])
        for c in ctxs:
            c.close()
    for i, sc in enumerate(scenes):
        To, no, _ = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(a0["Tc"][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")
        assert a0["nt"][i] == no


@pytest.mark.gpu
def test_streamed_batch_padded_rows_ragged_counts_and_bad_arguments(gpu_ctx, oracle):
    """Host images with padded rows (row_stride > width: one 2-D copy per image), per-pair live feature counts, a chunk
    size larger than the batch, 5 contexts for 3 pairs; and the argument checks (done before any thread starts)."""
    import copy
    W, Hh, L, N = 320, 240, 3, 150
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=4100 + i, margin=12) for i in range(3)]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    a, _, _ = _host_batch(scenes, L, W, Hh)
    RS = W + 24
    fr = np.full((3, Hh, RS), 7, np.uint8); fc = np.full((3, Hh, RS), 9, np.uint8)
    for i, sc in enumerate(scenes):
        fr[i, :, :W] = sc.ref_pyr[0]; fc[i, :, :W] = sc.cur_pyr[0]
    nf = np.array([150, 90, 10], np.int32)
    s = _stream_desc(a, fr, fc, 3, N, L, W, Hh, row_stride=RS, image_pitch=RS * Hh)
    s.n_features = nf.ctypes.data
    ctxs = [capi.Context(0) for _ in range(5)]
    arr = (C.c_void_p * 5)(*[c.handle for c in ctxs])
    assert gpu_ctx.lib.dsdtm_sparse_align_batch_streamed(arr, 5, C.byref(s), 1000, C.byref(cam), C.byref(prm)) == 0
    for i, sc in enumerate(scenes):
        s2 = copy.copy(sc)
        n = int(nf[i])
        s2.px, s2.bearing, s2.p_world, s2.initial = sc.px[:n], sc.bearing[:n], sc.p_world[:n], sc.initial[:n]
        To, no, _ = oracle.sparse_align(s2, L, 0, 10)
        assert a["nt"][i] == no
        H.assert_pose_close(a["Tc"][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")
    assert a["nt"][2] == 0 and np.array_equal(a["Tc"][2], scenes[2].T_cur_w_seed.reshape(12))
    # an empty batch is a no-op; a one-level "pyramid" (nothing to build on the device) works
    empty = capi.StreamDesc.from_buffer_copy(bytes(s))
    empty.n_pairs = 0
    assert gpu_ctx.lib.dsdtm_sparse_align_batch_streamed(arr, 5, C.byref(empty), 4, C.byref(cam), C.byref(prm)) == 0
    a1, _, _ = _host_batch(scenes, L, W, Hh)
    s1 = _stream_desc(a1, fr, fc, 3, N, 1, W, Hh, row_stride=RS, image_pitch=RS * Hh)
    prm1 = capi.AlignParams(1, 0, 10, 15)
    assert gpu_ctx.lib.dsdtm_sparse_align_batch_streamed(arr, 2, C.byref(s1), 2, C.byref(cam), C.byref(prm1)) == 0
    for i, sc in enumerate(scenes):
        To, no, _ = oracle.sparse_align(sc, 1, 0, 10)
        assert a1["nt"][i] == no
        H.assert_pose_close(a1["Tc"][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"one level, pair {i}")
    # refused up front: negative feature count, no levels, rows shorter than the image, a context listed twice
    for field, value in (("max_features", -1), ("levels", 0), ("row_stride", W - 1), ("image_pitch", 10)):
        bad = capi.StreamDesc.from_buffer_copy(bytes(s))
        setattr(bad, field, value)
        assert gpu_ctx.lib.dsdtm_sparse_align_batch_streamed(arr, 5, C.byref(bad), 4, C.byref(cam), C.byref(prm)) == capi.ERR_INVALID
    twice = (C.c_void_p * 2)(ctxs[0].handle, ctxs[0].handle)
    assert gpu_ctx.lib.dsdtm_sparse_align_batch_streamed(twice, 2, C.byref(s), 4, C.byref(cam), C.byref(prm)) == capi.ERR_INVALID
    # the sharded entry checks its descriptor before it sizes any staging from it (round-3 advice)
    _, b, _ = _host_batch(scenes, L, W, Hh)
    for field, value in (("max_features", -1), ("levels", 0), ("pyr_pitch", 0), ("pyr_pitch", 6)):
        bad = capi.BatchDesc.from_buffer_copy(bytes(b))
        setattr(bad, field, value)
        assert gpu_ctx.lib.dsdtm_sparse_align_batch_sharded(arr, 5, C.byref(bad), C.byref(cam), C.byref(prm)) == capi.ERR_INVALID
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_sharded_large_pairs_take_the_big_lds_kernels_on_every_context(gpu_ctx, oracle):
    """1000- and 2000-feature pairs run kernels that need the > 64 KB dynamic-LDS opt-in — per kernel AND per device
    (round 3 set it once per process). Three contexts, each launching from its own host thread: every shard must get a
    launchable kernel, and the oracle's results."""
    W, Hh, L = 320, 240, 3
    for N, P in ((1000, 99), (1900, 54)):      # 33 / 18 pairs per shard: too many for teams -> the one-CU / two-member LDS kernels
        base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=5200 + N + i, margin=12) for i in range(2)]
        want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
        scenes = [base[i % 2] for i in range(P)]
        cam = capi.camera_struct(scenes[0].cam)
        prm = capi.AlignParams(L, 0, 10, 15)
        a, b, _ = _host_batch(scenes, L, W, Hh)
        ctxs = [capi.Context(0) for _ in range(3)]
        arr = (C.c_void_p * 3)(*[c.handle for c in ctxs])
        rc = gpu_ctx.lib.dsdtm_sparse_align_batch_sharded(arr, 3, C.byref(b), C.byref(cam), C.byref(prm))
        assert rc == 0, [c.lib.dsdtm_last_error(c.handle) for c in ctxs]
        for c in ctxs:
            c.close()
        for i in range(P):
            To, no, _ = want[i % 2]
            H.assert_pose_close(a["Tc"][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"N={N} pair {i}")
            assert a["nt"][i] == no


@pytest.mark.gpu
@pytest.mark.diag
@pytest.mark.parametrize("N,P", [(600, 2), (1900, 18)])
def test_sharded_entry_fetches_the_results_again_after_a_transparent_rerun(oracle, N, P):
    """dsdtm_sparse_align_batch_sharded queues its downloads before it checks the launch. When that check re-seeds and re-runs a
    team / two-member launch whose wait for a partner workgroup ran out, the host arrays hold the aborted launch's poses: they
    must be fetched again (round-4 advice). The debug switch keeps a member of every pair away, so the first attempt always
    times out; the caller must still see DSDTM_OK and the oracle's results."""
    W, Hh, L = 320, 240, 3
    base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1300 + N + i, margin=12) for i in range(min(P, 3))]
    scenes = [base[i % len(base)] for i in range(P)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
    ctx = capi.Context(0, diag=True)                                   # the diagnostic library: a member of every pair stays away
    lib = ctx.lib
    drop = lib.dsdtm_debug_drop_team_members
    drop.restype, drop.argtypes = None, [C.c_int]
    rec = lib.dsdtm_debug_recovered_launches
    rec.restype, rec.argtypes = C.c_longlong, [C.c_void_p]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    a, b, _ = _host_batch(scenes, L, W, Hh)
    arr = (C.c_void_p * 1)(ctx.handle)
    n0 = rec(ctx.handle)
    try:
        drop(1)
        rc = lib.dsdtm_sparse_align_batch_sharded(arr, 1, C.byref(b), C.byref(cam), C.byref(prm))
    finally:
        drop(0)
    assert rc == 0, lib.dsdtm_last_error(ctx.handle)
    assert rec(ctx.handle) == n0 + 1                                   # the launch did time out and was re-run
    for i in range(P):
        To, no, so = want[i % len(base)]
        H.assert_pose_close(a["Tc"][i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i} after the re-run")
        assert a["nt"][i] == no and list(a["st"]["iters"][i][:L]) == list(so["iters"][:L])
    ctx.close()
