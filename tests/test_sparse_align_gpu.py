"""GPU parity of Sprase_ImgAlign::Run (HIP path through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from dsdtm_amd import synth
from tests import helpers as H
from tests.conftest import cached_scene

pytestmark = pytest.mark.gpu


@pytest.mark.diag
def test_device_building_blocks(gpu_ctx_diag, oracle):
    """DPP wave reduction, pivoted LDLT, SE(3) exp/mul on the device vs the oracle's."""
    gpu_ctx = gpu_ctx_diag            # the diagnostic library: this test needs its dsdtm_debug_* entries
    import ctypes as C
    rng = np.random.default_rng(7)
    lib = oracle.load()
    cases = []
    for i in range(24):
        J = rng.standard_normal((40, 6)) * np.array([30, 30, 30, 80, 80, 80])
        Hm = J.T @ J
        if i % 6 == 3:
            Hm[:, 2] = 0; Hm[2, :] = 0           # rank deficient
        if i % 6 == 4:
            Hm[:] = 0                              # zero matrix (no visible patch)
        if i % 6 == 5:
            Hm[3, 3] *= 1e6                        # pivoting order changes
        b = rng.standard_normal(6) * 1e-2
        xi = rng.standard_normal(6) * (1e-12 if i % 5 == 0 else (0.02 if i % 5 in (1, 2) else 0.3))
        cases.append(np.concatenate([H.upper21(Hm), b, xi]))
    cases = np.array(cases)
    out = H.selftest(gpu_ctx, cases)
    dp = C.POINTER(C.c_double)
    for c, o in zip(cases, out):
        Hm = np.zeros((6, 6))
        q = 0
        for i in range(6):
            for j in range(i, 6):
                Hm[i, j] = Hm[j, i] = c[q]; q += 1
        x = np.zeros(6)
        Hf = np.ascontiguousarray(Hm.reshape(36))
        lib.oracle_ldlt6_solve(Hf.ctypes.data_as(dp), np.ascontiguousarray(c[21:27]).ctypes.data_as(dp), x.ctypes.data_as(dp))
        scale = max(1e-300, np.abs(x).max())
        assert np.allclose(o[:6], x, rtol=1e-9, atol=1e-9 * scale), (o[:6], x)
        E = oracle.OracleSE3()
        lib.oracle_se3_exp(np.ascontiguousarray(c[27:33]).ctypes.data_as(dp), C.byref(E))
        assert np.allclose(o[6:10], list(E.q), atol=1e-14)
        assert np.allclose(o[10:13], list(E.t), atol=1e-14)
        E2, Eb = oracle.OracleSE3(), oracle.OracleSE3()
        lib.oracle_se3_exp(np.ascontiguousarray(c[21:27]).ctypes.data_as(dp), C.byref(Eb))
        lib.oracle_se3_mul(C.byref(E), C.byref(Eb), C.byref(E2))
        assert np.allclose(o[13:17], list(E2.q), atol=1e-14)
        assert np.allclose(o[17:20], list(E2.t), atol=1e-14)
        want = c[27] * (64 * 65 / 2)
        assert abs(o[20] - want) <= 1e-12 * abs(want) + 1e-300
        assert abs(o[21] - want) <= 1e-12 * abs(want) + 1e-300
        T = np.zeros(12)
        lib.oracle_se3_to_rt(C.byref(E), T.ctypes.data_as(dp))
        assert np.allclose(o[22:34], T, atol=1e-13)
        # H^+ by the lane-parallel Gauss-Jordan inversion: taken for every positive definite matrix, declined for
        # the degenerate ones, and equal to the pivoted LDLT's solves of the unit vectors to cond(H) * eps
        full_rank = np.linalg.matrix_rank(Hm) == 6
        assert bool(o[106]) == full_rank
        if full_rank:
            ref_inv = o[34:70]
            assert np.allclose(o[70:106], ref_inv, rtol=1e-8, atol=1e-9 * np.abs(ref_inv).max())
            assert np.allclose(o[70:106].reshape(6, 6).T @ Hm, np.eye(6), atol=1e-8)
        else:
            assert (o[70:106] == -12345.0).all()
        assert o[119] < 1e-13                                     # packed row reduction == eight separate row sums (to rounding)
        # the step in matrix form (what the solver publishes first) is the same group element as SE3::exp
        if np.dot(c[30:33], c[30:33]) < 0.01:
            assert np.allclose(o[107:116].reshape(3, 3), T.reshape(3, 4)[:, :3], atol=1e-15)
            assert np.allclose(o[116:119], T.reshape(3, 4)[:, 3], atol=1e-15 + 1e-15 * np.abs(T).max())


def test_config2_pose_matches_oracle(gpu_ctx, oracle):
    """BASELINE config 2: 640x480, 4 levels, 300 patches, cap 10."""
    sc = cached_scene()
    To, no, so = oracle.sparse_align(sc, 4, 0, 10)
    Tg, ng, sg = H.gpu_sparse_align(sc, 4, 0, 10, ctx=gpu_ctx)
    ang, dt = H.assert_pose_close(Tg, To, what="config2")
    assert ng == no
    assert sg["iters"] == so["iters"], (sg["iters"], so["iters"])
    assert sg["exit_code"] == so["exit_code"]
    assert sg["n_ref"] == so["n_ref"] and sg["n_vis"] == so["n_vis"]
    assert np.allclose(sg["chi2"], so["chi2"], rtol=1e-9)
    # FP64 decision parity: only summation-order noise remains
    assert ang <= H.TIGHT_RAD and dt <= H.TIGHT_M, (ang, dt)
    # and the result is right in absolute terms (ground truth of the synthetic scene)
    ea, et = synth.pose_error(Tg, sc.T_cur_w_true)
    assert ea < 2e-4 and et < 3e-4


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_random_scenes_match_oracle(gpu_ctx, oracle, seed):
    rng = np.random.default_rng(seed)
    sc = cached_scene(width=320, height=240, levels=3, n_patches=int(rng.integers(40, 320)), seed=100 + seed,
                      xi=tuple(synth.random_xi(rng)), depth=float(rng.uniform(1, 4)),
                      T_ref_w=tuple(map(tuple, synth.random_pose(rng))), margin=12,
                      frac_uninitial=0.1 if seed % 2 else 0.0)
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=gpu_ctx)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"seed {seed}")
    assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"]
    assert sg["n_ref"] == so["n_ref"] and sg["n_vis"] == so["n_vis"]


@pytest.mark.parametrize("n_patches,size", [(330, (320, 240)), (448, (320, 240)), (449, (320, 240)), (600, (640, 480)), (704, (640, 480)),
                                             (705, (640, 480)), (1000, (640, 480)), (2000, (640, 480)), (4096, (640, 480)), (4100, (640, 480))])
def test_large_patch_counts(gpu_ctx, oracle, n_patches, size):
    """448- and 704-lane register kernels, teams of 3..16 compute units (705..4096 features: configs 3 and 5 patch
    counts) and the workspace kernel beyond."""
    sc = cached_scene(width=size[0], height=size[1], levels=3, n_patches=n_patches, seed=77, margin=12)
    To, no, so = oracle.sparse_align(sc, 3, 0, 8)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 8, ctx=gpu_ctx)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"N={n_patches}")
    assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"]


def test_level_range_and_tracking_params(gpu_ctx, oracle):
    """Tracking's constructor arguments (5 levels, min 0, 8 iterations; src/Tracking.cpp:20-24,37)
    and a partial level range."""
    sc = cached_scene(width=640, height=480, levels=5, n_patches=200, seed=9)
    for (mx, mn, it) in [(5, 0, 8), (4, 2, 30), (3, 1, 1)]:
        To, no, so = oracle.sparse_align(sc, mx, mn, it)
        Tg, ng, sg = H.gpu_sparse_align(sc, mx, mn, it, ctx=gpu_ctx)
        H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=str((mx, mn, it)))
        assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"]


def test_too_few_features_returns_zero_and_keeps_pose(gpu_ctx, oracle):
    """Run(): size < Camera.Min_fts -> 0, pose untouched (src/Sprase_ImageAlign.cpp:34-38)."""
    sc = cached_scene(width=320, height=240, levels=3, n_patches=10, seed=5, margin=12)
    seed_pose = sc.T_cur_w_seed.copy()
    Tg, ng, _ = H.gpu_sparse_align(sc, 3, 0, 10, min_fts=15, ctx=gpu_ctx)
    To, no, _ = oracle.sparse_align(sc, 3, 0, 10, min_fts=15)
    assert ng == 0 and no == 0
    assert np.array_equal(Tg, seed_pose) and np.array_equal(To, seed_pose)


def test_no_visible_patch_leaves_pose(gpu_ctx, oracle):
    """Quirk Q11: every patch projects outside -> chi2 NaN, H = 0 -> step 0 -> pose unchanged."""
    import copy
    sc = copy.deepcopy(cached_scene(width=320, height=240, levels=3, n_patches=60, seed=6, margin=12))
    T_seed = synth.se3_exp([0, 0, 0, 0, 1.2, 0])[:3] @ np.vstack([sc.T_ref_w, [0, 0, 0, 1]])   # looks away
    To, no, so = oracle.sparse_align(sc, 3, 0, 10, T_seed=T_seed)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, T_seed=T_seed, ctx=gpu_ctx)
    assert no == 0 and ng == 0
    H.assert_pose_close(Tg, To, 1e-12, 1e-12, what="no visible")
    assert sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"]


def test_uninitial_zero_points_and_border_features(gpu_ctx, oracle):
    """Quirk Q3: skip !mbInitial, P_w exactly zero, and features within 3 px of the level border."""
    import copy
    sc = copy.deepcopy(cached_scene(width=320, height=240, levels=3, n_patches=120, seed=8, margin=12))
    sc.initial[::7] = 0
    sc.p_world[3::11] = 0.0
    sc.px[5] = (2.0, 100.0); sc.px[6] = (318.5, 100.0); sc.px[9] = (100.0, 237.2); sc.px[10] = (11.9, 12.1)
    sc.bearing = synth.bearing_from_px(sc.cam, sc.px)
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=gpu_ctx)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what="masking")
    assert ng == no and sg["n_ref"] == so["n_ref"] and sg["n_vis"] == so["n_vis"]
    assert sg["iters"] == so["iters"]


@pytest.mark.parametrize("seed,n,size,levels", [(31, 300, (640, 480), 4), (32, 200, (320, 240), 3), (33, 1000, (640, 480), 4)])
def test_occluders_moving_objects_and_exposure_change(gpu_ctx, oracle, seed, n, size, levels):
    """Inputs the synthetic plane does not have: the current frame with occluding blocks (a fifth of the patches land on them),
    objects that moved on their own between the frames, saturated blobs and a global exposure change. The reference has no
    robust weights (src/Sprase_ImageAlign.cpp:282-291: plain squared residuals), so these are simply large residuals that
    drag the solution — the same solution, iteration for iteration, in the HIP path and the oracle."""
    import copy
    W, Hh = size
    sc = copy.deepcopy(cached_scene(width=W, height=Hh, levels=levels, n_patches=n, seed=seed, margin=20))
    rng = np.random.default_rng(seed)
    cur = sc.cur_pyr[0].astype(np.int32)
    ref = sc.ref_pyr[0]
    for _ in range(12):                                        # occluders: flat and textured blocks
        w, h = int(rng.integers(W // 20, W // 6)), int(rng.integers(Hh // 20, Hh // 6))
        x, y = int(rng.integers(0, W - w)), int(rng.integers(0, Hh - h))
        cur[y:y + h, x:x + w] = rng.integers(0, 256) if rng.random() < 0.5 else rng.integers(0, 256, (h, w))
    for _ in range(6):                                         # moving objects: a piece of the reference pasted 5-15 px away
        w, h = int(rng.integers(30, 80)), int(rng.integers(30, 80))
        x, y = int(rng.integers(20, W - w - 20)), int(rng.integers(20, Hh - h - 20))
        dx, dy = int(rng.integers(-15, 16)), int(rng.integers(-15, 16))
        cur[y + dy:y + dy + h, x + dx:x + dx + w] = ref[y:y + h, x:x + w]
    for _ in range(4):                                         # saturated blobs
        x, y, r = int(rng.integers(40, W - 40)), int(rng.integers(40, Hh - 40)), int(rng.integers(8, 30))
        yy, xx = np.mgrid[0:Hh, 0:W]
        cur[(xx - x) ** 2 + (yy - y) ** 2 <= r * r] = 255
    cur = np.clip(cur * 1.12 + 9, 0, 255).astype(np.uint8)    # exposure change
    sc.cur_pyr = synth.build_pyramid(cur, levels)
    To, no, so = oracle.sparse_align(sc, levels, 0, 10)
    Tg, ng, sg = H.gpu_sparse_align(sc, levels, 0, 10, ctx=gpu_ctx)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what="occluded scene")
    assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"] and sg["n_vis"] == so["n_vis"]
    ea, et = synth.pose_error(To, sc.T_cur_w_true)
    assert ea > 1e-4 or et > 1e-4                              # the outliers do drag the estimate: this is not the clean scene again


def test_identity_motion_converges_immediately(gpu_ctx, oracle):
    sc = cached_scene(width=320, height=240, levels=3, n_patches=150, seed=11, xi=(0, 0, 0, 0, 0, 0), margin=12)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=gpu_ctx)
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10)
    ea, et = synth.pose_error(Tg, sc.T_cur_w_true)
    assert ea < 1e-4 and et < 2e-4 and sg["iters"] == so["iters"]


@pytest.mark.parametrize("name", ["sparse_align_a.npz", "sparse_align_b.npz"])
def test_golden_fixtures(gpu_ctx, name):
    """HIP path against the committed golden vectors (tests/golden, generated by make_golden.py)."""
    g = H.GoldenScene(H.golden_path(name))
    mx, mn, it, mf = g.params
    Tg, ng, sg = H.gpu_sparse_align(g, mx, mn, it, min_fts=mf, ctx=gpu_ctx)
    H.assert_pose_close(Tg, g.d["out_T"], H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=name)
    assert ng == int(g.d["out_n"])
    assert sg["iters"] == list(g.d["out_iters"]) and sg["exit_code"] == list(g.d["out_exit"])
    assert sg["n_ref"] == list(g.d["out_nref"]) and sg["n_vis"] == list(g.d["out_nvis"])


@pytest.mark.parametrize("NMAX,base_counts,reps", [(300, [300, 257, 64, 15, 14, 0, 129], 1),
                                                   (1000, [1000, 705, 257, 64, 15, 14, 0, 999, 449], 8)])
def test_batch_device_api_with_ragged_feature_counts(gpu_ctx, oracle, NMAX, base_counts, reps):
    """dsdtm_sparse_align_batch_device: several pairs in one launch, per-pair feature counts
    (including one below Camera.Min_fts and one empty), stats array, in/out pose buffer. Register kernel
    (300 features at most) and, with 72 pairs of up to 1000, the workspace kernel with its per-patch windows."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    counts = base_counts * reps                      # (first case) odd pair count: the last workgroup has one empty pair slot
    base_scenes = []
    for i, n in enumerate(base_counts):
        rng = np.random.default_rng(500 + i)
        base_scenes.append(cached_scene(width=W, height=Hh, levels=L, n_patches=max(n, 1), seed=600 + i, margin=12,
                                        xi=tuple(synth.random_xi(rng)), depth=float(rng.uniform(1, 4)),
                                        T_ref_w=tuple(map(tuple, synth.random_pose(rng)))))
    scenes = base_scenes * reps
    ws, hs, st, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    pitch = (nbytes + 255) // 256 * 256
    P = len(counts)
    ref = np.zeros((P, pitch), np.uint8); cur = np.zeros((P, pitch), np.uint8)
    px = np.zeros((P, NMAX, 2), np.float32); bear = np.zeros((P, NMAX, 3)); pw = np.zeros((P, NMAX, 3))
    ini = np.zeros((P, NMAX), np.uint8); Tr = np.zeros((P, 12)); Tc = np.zeros((P, 12))
    for i, (sc, n) in enumerate(zip(scenes, counts)):
        for l in range(L):
            ref[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.ref_pyr[l].reshape(-1)
            cur[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.cur_pyr[l].reshape(-1)
        px[i, :n] = sc.px[:n]; bear[i, :n] = sc.bearing[:n]; pw[i, :n] = sc.p_world[:n]; ini[i, :n] = sc.initial[:n]
        # poison the padding: it must never be read
        px[i, n:] = np.nan; bear[i, n:] = np.nan; pw[i, n:] = np.nan; ini[i, n:] = 1
        Tr[i] = sc.T_ref_w.reshape(12); Tc[i] = sc.T_cur_w_seed.reshape(12)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(ref=ref, cur=cur, px=px, bear=bear, pw=pw, ini=ini, Tr=Tr, Tc=Tc).items()}
    t["nf"] = torch.tensor(counts, dtype=torch.int32, device=dev)
    t["nt"] = torch.full((P,), -7, dtype=torch.int32, device=dev)
    t["st"] = torch.zeros((P, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = P, NMAX, L
    for l in range(L):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], st[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (t[k].data_ptr() for k in ("ref", "cur", "px", "bear", "pw"))
    b.initial, b.n_features, b.T_ref_w, b.T_cur_w = t["ini"].data_ptr(), t["nf"].data_ptr(), t["Tr"].data_ptr(), t["Tc"].data_ptr()
    b.n_tracked, b.stats = t["nt"].data_ptr(), t["st"].data_ptr()
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    stream = torch.cuda.Stream(device=dev)
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm),
                                                              stream.cuda_stream))
    stream.synchronize()
    Tg = t["Tc"].cpu().numpy(); ntg = t["nt"].cpu().numpy()
    stg = np.frombuffer(t["st"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
    import copy
    want = []
    for sc, n in zip(base_scenes, base_counts):
        s2 = copy.copy(sc)
        s2.px, s2.bearing, s2.p_world, s2.initial = sc.px[:n], sc.bearing[:n], sc.p_world[:n], sc.initial[:n]
        want.append(oracle.sparse_align(s2, L, 0, 10))
    for i, (sc, n) in enumerate(zip(scenes, counts)):
        To, no, so = want[i % len(base_counts)]
        assert ntg[i] == no, (i, ntg[i], no)
        if n < 15:
            assert no == 0 and np.array_equal(Tg[i], sc.T_cur_w_seed.reshape(12))      # pose untouched
        else:
            H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")
            assert list(stg["iters"][i]) == so["iters"] and list(stg["n_ref"][i]) == so["n_ref"]


def _device_batch(torch, dev, scenes, L, W, Hh):
    """Packs scenes into device tensors + a BatchDesc (rows contiguous, all features used)."""
    from dsdtm_amd import capi
    ws, hs, st, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    pitch = (nbytes + 255) // 256 * 256
    P, N = len(scenes), len(scenes[0].px)
    ref = np.zeros((P, pitch), np.uint8); cur = np.zeros((P, pitch), np.uint8)
    for i, sc in enumerate(scenes):
        for l in range(L):
            ref[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.ref_pyr[l].reshape(-1)
            cur[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.cur_pyr[l].reshape(-1)
    arr = dict(ref=ref, cur=cur, px=np.stack([s.px for s in scenes]), bear=np.stack([s.bearing for s in scenes]),
               pw=np.stack([s.p_world for s in scenes]), ini=np.stack([s.initial for s in scenes]),
               Tr=np.stack([s.T_ref_w.reshape(12) for s in scenes]), Tc=np.stack([s.T_cur_w_seed.reshape(12) for s in scenes]))
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in arr.items()}
    t["nt"] = torch.zeros(P, dtype=torch.int32, device=dev)
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = P, N, L
    for l in range(L):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], st[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (t[k].data_ptr() for k in ("ref", "cur", "px", "bear", "pw"))
    b.initial, b.n_features, b.T_ref_w, b.T_cur_w = t["ini"].data_ptr(), None, t["Tr"].data_ptr(), t["Tc"].data_ptr()
    b.n_tracked, b.stats = t["nt"].data_ptr(), None
    return t, b


def test_launches_in_flight_on_two_streams(gpu_ctx, oracle):
    """Two batches launched back to back on two streams of one context (each launch has its own pair
    counter; the persistent workgroups of both are resident together): both give the oracle's results."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    groups = [[cached_scene(width=W, height=Hh, levels=L, n_patches=n, seed=800 + 10 * g + i, margin=12) for i in range(p)]
              for g, (n, p) in enumerate([(150, 9), (300, 7)])]
    streams = [torch.cuda.Stream(device=dev) for _ in groups]
    packed = [_device_batch(torch, dev, scenes, L, W, Hh) for scenes in groups]
    cam = capi.camera_struct(groups[0][0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    torch.cuda.synchronize()
    for rep in range(3):                                   # several rounds: counters are recycled from a ring
        for (t, b), scenes in zip(packed, groups):
            t["Tc"].copy_(torch.from_numpy(np.stack([s.T_cur_w_seed.reshape(12) for s in scenes])).to(dev))
        torch.cuda.synchronize()
        for (t, b), st in zip(packed, streams):
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
        for st in streams:
            st.synchronize()
        for (t, b), scenes in zip(packed, groups):
            Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
            for i, sc in enumerate(scenes):
                To, no, _ = oracle.sparse_align(sc, L, 0, 10)
                H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"round {rep} pair {i}")
                assert ntg[i] == no


def test_repeated_launches_are_bitwise_deterministic(gpu_ctx):
    """The pair-local LDS hand-over protocol (two pairs per workgroup, speculative factorisation)
    must not depend on timing: identical inputs -> bit-identical poses, launch after launch, for
    full and for partially filled waves."""
    for kw in (dict(width=320, height=240, levels=3, n_patches=140, seed=77, margin=12, frac_uninitial=0.05),
               dict(width=640, height=480, levels=4, n_patches=300, seed=0xD5D7)):
        sc = cached_scene(**kw)
        L = kw["levels"]
        ref = [H.gpu_sparse_align(sc, L, 0, 10, ctx=gpu_ctx) for _ in range(6)]
        for T, n, st in ref[1:]:
            assert np.array_equal(T, ref[0][0]) and n == ref[0][1] and st["iters"] == ref[0][2]["iters"]
            assert st["chi2"] == ref[0][2]["chi2"]


def test_device_api_with_padded_rows(gpu_ctx, oracle):
    """Device pyramids whose rows are `stride` > width bytes apart. The kernels index rows with
    `stride` on both sides; the reference mixes `cols` and `step` on the current image (quirk Q7,
    src/Sprase_ImageAlign.cpp:272-281), which is only self-consistent for continuous cv::Mats — so the
    expected result is the oracle's on contiguous copies of the same images."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    L, N = 3, 180
    scenes = [cached_scene(width=320, height=240, levels=L, n_patches=N, seed=900 + i, margin=12) for i in range(3)]
    ws, hs = [320, 160, 80], [240, 120, 60]
    st = [w + 12 for w in ws]
    offs, off = [], 0
    for l in range(L):
        offs.append(off)
        off += (st[l] * hs[l] + 63) // 64 * 64
    pitch = (off + 255) // 256 * 256
    P = len(scenes)
    ref = np.full((P, pitch), 0xAB, np.uint8); cur = np.full((P, pitch), 0xCD, np.uint8)     # poison the padding
    px = np.zeros((P, N, 2), np.float32); bear = np.zeros((P, N, 3)); pw = np.zeros((P, N, 3))
    ini = np.zeros((P, N), np.uint8); Tr = np.zeros((P, 12)); Tc = np.zeros((P, 12))
    for i, sc in enumerate(scenes):
        for l in range(L):
            for buf, pyr in ((ref, sc.ref_pyr), (cur, sc.cur_pyr)):
                view = buf[i, offs[l]:offs[l] + st[l] * hs[l]].reshape(hs[l], st[l])
                view[:, :ws[l]] = pyr[l]
        px[i], bear[i], pw[i], ini[i] = sc.px, sc.bearing, sc.p_world, sc.initial
        Tr[i] = sc.T_ref_w.reshape(12); Tc[i] = sc.T_cur_w_seed.reshape(12)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(ref=ref, cur=cur, px=px, bear=bear, pw=pw, ini=ini, Tr=Tr, Tc=Tc).items()}
    t["nt"] = torch.zeros(P, dtype=torch.int32, device=dev)
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = P, N, L
    for l in range(L):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], st[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (t[k].data_ptr() for k in ("ref", "cur", "px", "bear", "pw"))
    b.initial, b.n_features, b.T_ref_w, b.T_cur_w = t["ini"].data_ptr(), None, t["Tr"].data_ptr(), t["Tc"].data_ptr()
    b.n_tracked, b.stats = t["nt"].data_ptr(), None
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    stream = torch.cuda.Stream(device=dev)
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), stream.cuda_stream))
    stream.synchronize()
    Tg = t["Tc"].cpu().numpy(); ntg = t["nt"].cpu().numpy()
    for i, sc in enumerate(scenes):
        To, no, _ = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"strided pair {i}")
        assert ntg[i] == no


@pytest.mark.parametrize("seed", range(12))
def test_randomised_configurations(gpu_ctx, oracle, seed):
    """Random image sizes (odd and even), level ranges, feature counts across the kernel shapes,
    iteration caps, reference poses, motions and shares of uninitialised features."""
    rng = np.random.default_rng(1000 + seed)
    width, height = int(rng.integers(150, 420)), int(rng.integers(120, 330))
    levels = int(rng.integers(2, 6))
    while min(width, height) >> (levels - 1) < 24:
        levels -= 1
    n = int(rng.choice([18, 64, 127, 130, 190, 200, 256, 300, 321, 450, 520]))
    max_level = int(rng.integers(1, levels + 1)); min_level = int(rng.integers(0, max_level))
    iters = int(rng.integers(1, 13))
    xi = tuple(rng.uniform(-1, 1, 6) * np.array([0.012, 0.012, 0.012, 0.006, 0.006, 0.006]) * rng.uniform(0.2, 2.0))
    sc = synth.make_scene(width=width, height=height, levels=levels, n_patches=n, seed=2000 + seed, xi=xi, margin=8,
                          T_ref_w=synth.random_pose(rng), frac_uninitial=float(rng.choice([0.0, 0.1, 0.4])),
                          depth=float(rng.uniform(1.0, 5.0)))
    To, no, so = oracle.sparse_align(sc, max_level, min_level, iters)
    Tg, ng, sg = H.gpu_sparse_align(sc, max_level, min_level, iters, ctx=gpu_ctx)
    what = f"seed {seed}: {width}x{height} L{levels} [{min_level},{max_level}) n={n} it={iters}"
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=what)
    assert ng == no, what
    for k in ("iters", "exit_code", "n_ref", "n_vis"):
        assert sg[k] == so[k], (what, k)


@pytest.mark.parametrize("scale", [1.0, 2.5])
def test_window_refills_under_multi_pixel_motion(gpu_ctx, oracle, scale):
    """The kernel keeps each patch's current-image footprint in an LDS window and refills a lane's
    window when its floor position leaves it. Fine levels only, started from the identity seed: the
    patches move by several pixels between the first iterations, so most lanes refill repeatedly."""
    xi = tuple(scale * v for v in (0.01, -0.006, 0.004, 0.004, -0.003, 0.005))
    sc = cached_scene(width=640, height=480, levels=3, n_patches=300, seed=77, xi=xi, margin=40)
    for (mx, mn, it) in [(1, 0, 30), (2, 0, 30), (2, 1, 12)]:
        To, no, so = oracle.sparse_align(sc, mx, mn, it)
        Tg, ng, sg = H.gpu_sparse_align(sc, mx, mn, it, ctx=gpu_ctx)
        H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"scale {scale} levels {(mx, mn)}")
        assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"] and sg["n_vis"] == so["n_vis"]


def test_footprints_at_the_end_of_the_pyramid_allocation(gpu_ctx, oracle):
    """Tightly packed device pyramids (stride == width, pitch == end of the last level, no padding) with
    features along the right and bottom borders of every level: the 12-byte row gathers of the last
    rows would run past the allocation and take the shifted-window path; results must not change."""
    import copy
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    L, N = 3, 120
    ws, hs = [320, 160, 80], [240, 120, 60]
    scenes = []
    for i in range(2):
        sc = copy.deepcopy(cached_scene(width=320, height=240, levels=L, n_patches=N, seed=950 + i, margin=12))
        rng = np.random.default_rng(i)
        k = N // 3
        # bottom rows and right columns, just inside the coarsest level's 3-px border (x4 at level 0)
        sc.px[:k, 0] = rng.uniform(20, 300, k); sc.px[:k, 1] = rng.uniform(224.0, 227.9, k)
        sc.px[k:2 * k, 0] = rng.uniform(304.0, 307.9, k); sc.px[k:2 * k, 1] = rng.uniform(20, 220, k)
        sc.px[2 * k, :] = (307.5, 227.5)
        sc.px = sc.px.astype(np.float32)
        sc.bearing = synth.bearing_from_px(sc.cam, sc.px)
        X_r = sc.bearing * (sc.depth / sc.bearing[:, 2:3])
        Rr, tr = sc.T_ref_w[:, :3], sc.T_ref_w[:, 3]
        sc.p_world = (X_r - tr) @ Rr
        scenes.append(sc)
    offs, off = [], 0
    for l in range(L):
        offs.append(off)
        off += ws[l] * hs[l]
    pitch = off                                             # 100800: a multiple of 4, nothing behind the last pixel
    P = len(scenes)
    ref = np.zeros((P, pitch), np.uint8); cur = np.zeros((P, pitch), np.uint8)
    for i, sc in enumerate(scenes):
        for l in range(L):
            ref[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.ref_pyr[l].reshape(-1)
            cur[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.cur_pyr[l].reshape(-1)
    arr = dict(ref=ref, cur=cur, px=np.stack([s.px for s in scenes]), bear=np.stack([s.bearing for s in scenes]),
               pw=np.stack([s.p_world for s in scenes]), ini=np.stack([s.initial for s in scenes]),
               Tr=np.stack([s.T_ref_w.reshape(12) for s in scenes]), Tc=np.stack([s.T_cur_w_seed.reshape(12) for s in scenes]))
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in arr.items()}
    t["nt"] = torch.zeros(P, dtype=torch.int32, device=dev)
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = P, N, L
    for l in range(L):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], ws[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (t[k].data_ptr() for k in ("ref", "cur", "px", "bear", "pw"))
    b.initial, b.n_features, b.T_ref_w, b.T_cur_w = t["ini"].data_ptr(), None, t["Tr"].data_ptr(), t["Tc"].data_ptr()
    b.n_tracked, b.stats = t["nt"].data_ptr(), None
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    stream = torch.cuda.Stream(device=dev)
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), stream.cuda_stream))
    stream.synchronize()
    Tg = t["Tc"].cpu().numpy(); ntg = t["nt"].cpu().numpy()
    for i, sc in enumerate(scenes):
        To, no, _ = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"tight pair {i}")
        assert ntg[i] == no and no > 0


@pytest.mark.parametrize("n_patches", [16, 100, 128, 129, 192, 193, 256, 257, 320])
def test_every_register_kernel_shape(gpu_ctx, oracle, n_patches):
    """Feature counts on both sides of every kernel-shape boundary (2+1x4, 3+1x3, 4+1x2, 5+1x2, 7+1x1 waves)."""
    sc = cached_scene(width=320, height=240, levels=3, n_patches=n_patches, seed=300 + n_patches, margin=12)
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=gpu_ctx)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"N={n_patches}")
    assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"]
    assert sg["n_ref"] == so["n_ref"] and sg["n_vis"] == so["n_vis"]


def test_chained_sequence_like_tracking(gpu_ctx, oracle):
    """BASELINE config 3's flow at test size: a sequence of frames aligned against one reference,
    each result seeding the next Run (Test/test_SpraseImg_alignment.cpp:153-166,
    src/Tracking.cpp:201) — through the workspace kernel (N = 600 > 448)."""
    import copy
    rng = np.random.default_rng(31)
    base = cached_scene(width=320, height=240, levels=3, n_patches=600, seed=31, margin=12)
    tex = synth.make_texture(240, 320, 31)
    Tg = base.T_cur_w_seed.copy(); To = base.T_cur_w_seed.copy()
    xi = np.zeros(6)
    for k in range(6):
        xi = xi + np.concatenate([rng.uniform(-0.006, 0.006, 3), rng.uniform(-0.003, 0.003, 3)])
        sc = copy.copy(base)
        T_cr = synth.se3_exp(xi)
        sc.cur_pyr = synth.build_pyramid(synth.warp_plane(tex, base.cam, T_cr, base.depth), 3)
        To, no, so = oracle.sparse_align(sc, 3, 0, 8, T_seed=To)
        Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 8, T_seed=Tg, ctx=gpu_ctx)
        H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"frame {k}")
        assert ng == no and sg["iters"] == so["iters"]
        truth = (T_cr @ np.vstack([base.T_ref_w, [0, 0, 0, 1]]))[:3]
        ea, et = synth.pose_error(Tg, truth)
        assert ea < 5e-4 and et < 1e-3, (k, ea, et)


def test_device_resident_frames_track_a_sequence(gpu_ctx, oracle):
    """dsdtm_frame: a frame's pyramid is uploaded once and serves as `cur` of one Run and `ref` of the
    next (src/Tracking.cpp:201-224). Same kernel as the host entry point: identical results bit for
    bit; a frame built from level 0 alone (device pyrDown) gives the same pyramid, hence the same pose."""
    from dsdtm_amd import capi
    from dsdtm_amd.frame import Frame
    from dsdtm_amd.sparse_align import Sprase_ImgAlign
    rng = np.random.default_rng(77)
    base = cached_scene(width=320, height=240, levels=3, n_patches=150, seed=41, margin=12)
    tex = synth.make_texture(240, 320, 41)
    frames_h, frames_d = [], []
    xi = np.zeros(6)
    for k in range(4):
        pyr = synth.build_pyramid(synth.warp_plane(tex, base.cam, synth.se3_exp(xi), base.depth), 3) if k else base.ref_pyr
        for lst in (frames_h, frames_d):
            f = Frame(base.cam, pyr, base.T_ref_w)
            f.set_features(base.px, base.bearing, base.p_world, base.initial)   # the plane's points, as seen from frame 0
            lst.append(f)
        xi = xi + np.concatenate([rng.uniform(-0.004, 0.004, 3), rng.uniform(-0.002, 0.002, 3)])
    # every frame is aligned against frame 0 (whose features are valid), seeded with the previous pose
    al_h = Sprase_ImgAlign(3, 0, 8, ctx=gpu_ctx)
    al_d = Sprase_ImgAlign(3, 0, 8, ctx=gpu_ctx, resident_frames=True)
    for k in range(1, 4):
        for al, fr in ((al_h, frames_h), (al_d, frames_d)):
            fr[k].Set_Pose(fr[k - 1].Get_Pose())
            al.Run(fr[k], fr[0])
        assert np.array_equal(frames_h[k].Get_Pose(), frames_d[k].Get_Pose()), k
        assert al_h.last_stats == al_d.last_stats
    assert frames_d[0]._device_frame.handle is not None                      # uploaded once, reused three times
    # level 0 only + device pyrDown == the CPU-built pyramid (pyrDown is bit-exact)
    cur_img = frames_h[3].mvImg_Pyr[0]
    df = capi.DeviceFrame.from_image(gpu_ctx, cur_img, 3)
    f = Frame(base.cam, frames_h[3].mvImg_Pyr, frames_h[2].Get_Pose())
    f._device_frame = df
    al_d.Run(f, frames_d[0])
    assert np.array_equal(f.Get_Pose(), frames_h[3].Get_Pose())
    df.close()


@pytest.mark.parametrize("N,P", [(200, 5), (600, 3)])
def test_batch_launch_is_graph_capturable(gpu_ctx, oracle, N, P):
    """The device entry point only enqueues (one kernel; for teams a memset node before it): it can be
    captured into a hipGraph and replayed — no allocation, no synchronisation in the launch path, and the
    pair counter is back at zero after every replay. 200 features: the register kernel; 600 features on 3
    pairs: teams of 3 compute units (their exchange buffers are zeroed by a memset node of the graph)."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=700 + i, margin=12) for i in range(P)]
    ws, hs, st, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    pitch = (nbytes + 255) // 256 * 256
    ref = np.zeros((P, pitch), np.uint8); cur = np.zeros((P, pitch), np.uint8)
    for i, sc in enumerate(scenes):
        for l in range(L):
            ref[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.ref_pyr[l].reshape(-1)
            cur[i, offs[l]:offs[l] + ws[l] * hs[l]] = sc.cur_pyr[l].reshape(-1)
    arr = dict(ref=ref, cur=cur, px=np.stack([s.px for s in scenes]), bear=np.stack([s.bearing for s in scenes]),
               pw=np.stack([s.p_world for s in scenes]), ini=np.stack([s.initial for s in scenes]),
               Tr=np.stack([s.T_ref_w.reshape(12) for s in scenes]), seed=np.stack([s.T_cur_w_seed.reshape(12) for s in scenes]))
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in arr.items()}
    t["Tc"] = torch.zeros((P, 12), dtype=torch.float64, device=dev)
    t["nt"] = torch.zeros(P, dtype=torch.int32, device=dev)
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = P, N, L
    for l in range(L):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], st[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr, b.px_xy, b.bearing, b.p_world = (t[k].data_ptr() for k in ("ref", "cur", "px", "bear", "pw"))
    b.initial, b.n_features, b.T_ref_w, b.T_cur_w = t["ini"].data_ptr(), None, t["Tr"].data_ptr(), t["Tc"].data_ptr()
    b.n_tracked, b.stats = t["nt"].data_ptr(), None
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        t["Tc"].copy_(t["seed"])                                   # re-seed, then one launch: one "step"
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm),
                                                                  torch.cuda.current_stream().cuda_stream))
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in scenes]
    for rep in range(3):
        t["Tc"].zero_(); t["nt"].zero_()
        g.replay()
        torch.cuda.synchronize()
        Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
        for i, (To, no, _) in enumerate(want):
            H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"replay {rep} pair {i}")
            assert ntg[i] == no


def test_team_kernel_over_several_compute_units(gpu_ctx_each, oracle):
    """More features than one workgroup's registers hold (N > 704) and few pairs: every pair runs on a TEAM
    of K workgroups that exchange their partial sums and the pose through HBM each iteration. Same results
    as the oracle and — to rounding — as the single-workgroup workspace kernel (option no_team), bit-identical
    from launch to launch, ragged feature counts included; with many pairs the launcher falls back."""
    gpu_ctx = gpu_ctx_each            # the release library, then the diagnostic one (its switches only there)
    import ctypes as C
    import os
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N = 320, 240, 3, 1000
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=900 + i, margin=12) for i in range(3)]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in scenes]

    def run(t, b, scenes_):
        t["Tc"].copy_(torch.from_numpy(np.stack([s.T_cur_w_seed.reshape(12) for s in scenes_])).to(dev))
        torch.cuda.synchronize()
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), None))
        torch.cuda.synchronize()
        return t["Tc"].cpu().numpy().copy(), t["nt"].cpu().numpy().copy()

    t, b = _device_batch(torch, dev, scenes, L, W, Hh)
    T1, n1 = run(t, b, scenes)
    for i, (To, no, _) in enumerate(want):
        H.assert_pose_close(T1[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"team pair {i}")
        assert n1[i] == no
    for _ in range(3):                                        # no dependence on timing
        T2, n2 = run(t, b, scenes)
        assert np.array_equal(T1, T2) and np.array_equal(n1, n2)
    if gpu_ctx.diag:
        with capi.debug_options(no_team=1):
            Tw, nw = run(t, b, scenes)
        assert np.array_equal(n1, nw) and np.abs(T1 - Tw).max() < 1e-12
    # ragged feature counts: members without a live patch still take part in every exchange
    nf = np.array([1000, 720, 300], np.int32)
    t["nf"] = torch.from_numpy(nf).to(dev)
    b.n_features = t["nf"].data_ptr()
    T3, n3 = run(t, b, scenes)
    for i, sc in enumerate(scenes):
        sub = type("S", (), {})()
        for k in ("cam", "ref_pyr", "cur_pyr", "T_ref_w", "T_cur_w_seed"):
            setattr(sub, k, getattr(sc, k))
        sub.px, sub.bearing, sub.p_world, sub.initial = sc.px[:nf[i]], sc.bearing[:nf[i]], sc.p_world[:nf[i]], sc.initial[:nf[i]]
        To, no, _ = oracle.sparse_align(sub, L, 0, 10)
        H.assert_pose_close(T3[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"ragged pair {i}")
        assert n3[i] == no
    # two team launches in flight on two streams of one context (each has its own exchange buffers)
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    t2, b2 = _device_batch(torch, dev, scenes[::-1], L, W, Hh)
    b.n_features = None
    for tt, sc_ in ((t, scenes), (t2, scenes[::-1])):
        tt["Tc"].copy_(torch.from_numpy(np.stack([s_.T_cur_w_seed.reshape(12) for s_ in sc_])).to(dev))
    torch.cuda.synchronize()
    for bb, s_ in ((b, st[0]), (b2, st[1])):
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(bb), C.byref(cam), C.byref(prm), s_.cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(t["Tc"].cpu().numpy(), T1) and np.array_equal(t2["Tc"].cpu().numpy(), T1[::-1])
    # 20 pairs still run as teams (24 x 4 workgroups), 60 pairs go to the workspace kernel: same answers
    for P in (20, 60):
        many = [scenes[i % 3] for i in range(P)]
        tm, bm = _device_batch(torch, dev, many, L, W, Hh)
        Tm, nm = run(tm, bm, many)
        for i in range(P):
            assert nm[i] == want[i % 3][1]
            H.assert_pose_close(Tm[i], want[i % 3][0], H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"{P} pairs, pair {i}")


@pytest.mark.parametrize("N", [4096, 4097, 5000, 9000])
def test_large_teams_have_no_feature_count_cliff(gpu_ctx, oracle, N):
    """Up to 16 members (4096 features) a team runs the instantiation of exactly its size; 17..64 members
    (<= 16 384 features) run the 64-member instantiation with the member count at run time — 4097 features no
    longer fall to the single-CU workspace kernel (0.18 -> 0.66 ms in round 1). More than 32 members do not fit
    one XCD and are spread over all of them. Same results as the oracle either way."""
    sc = cached_scene(width=640, height=480, levels=4, n_patches=N, seed=4000 + N, margin=30)
    To, no, so = oracle.sparse_align(sc, 4, 0, 10)
    T1 = None
    for rep in range(2):
        Tg, ng, sg = H.gpu_sparse_align(sc, 4, 0, 10, ctx=gpu_ctx)
        H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"N={N}")
        assert ng == no and sg["iters"] == so["iters"] and sg["n_ref"] == so["n_ref"]
        if T1 is not None:
            assert np.array_equal(T1, Tg)                          # bit-identical from launch to launch
        T1 = Tg


def _short_team(ctx):
    import ctypes as C
    from dsdtm_amd import capi
    f = ctx.lib.dsdtm_debug_sparse_align_short_team
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(capi.BatchDesc), C.POINTER(capi.Camera), C.POINTER(capi.AlignParams), C.c_void_p]
    g = ctx.lib.dsdtm_debug_recovered_launches
    g.restype = C.c_longlong
    g.argtypes = [C.c_void_p]
    return f, g


@pytest.mark.diag
@pytest.mark.parametrize("N,P", [(600, 2), (1900, 18)])
def test_a_wait_that_runs_out_is_recovered_transparently(gpu_ctx_diag, oracle, N, P):
    """The multi-CU kernels' waits on partner workgroups are bounded. A debug entry launches teams (2 pairs of 600
    features) / two-member pairs (18 pairs of 1900 features: too many for teams) with a member missing, so the members that run can never complete an exchange:
    they must give up, stop the pair and let the kernel end — and dsdtm_sparse_align_check must re-seed the poses and
    re-run the batch on the one-CU kernels in the same call: DSDTM_OK, the oracle's results, like the reference's Run
    (src/Sprase_ImageAlign.cpp:29-60), which cannot fail for scheduling reasons."""
    gpu_ctx = gpu_ctx_diag            # the diagnostic library: this test needs its dsdtm_debug_* entries
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1300 + N + i, margin=12) for i in range(min(P, 3))]
    scenes = [base[i % len(base)] for i in range(P)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
    t, b = _device_batch(torch, dev, scenes, L, W, Hh)
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    f, recovered = _short_team(gpu_ctx)
    st = torch.cuda.Stream(device=dev)
    n0 = recovered(gpu_ctx.handle)
    gpu_ctx.check(f(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, st.cuda_stream))   # returns OK: the kernel ended, the batch was re-run
    assert recovered(gpu_ctx.handle) == n0 + 1
    Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
    for i in range(P):
        To, no, _ = want[i % len(base)]
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"re-run after the timeout, pair {i}")
        assert ntg[i] == no
    # nothing is left behind: an ordinary launch on the same stream is settled without a re-run
    t["Tc"].copy_(torch.from_numpy(np.stack([s.T_cur_w_seed.reshape(12) for s in scenes])).to(dev))
    torch.cuda.synchronize()
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, st.cuda_stream))
    assert recovered(gpu_ctx.handle) == n0 + 1
    assert np.array_equal(t["Tc"].cpu().numpy(), Tg) or np.allclose(t["Tc"].cpu().numpy(), Tg, atol=1e-12, rtol=0)


@pytest.mark.diag
def test_the_timeout_word_is_per_context(gpu_ctx_diag, oracle):
    """Two contexts on device 0; a timeout is provoked in one of them with the re-run switched off, so that it surfaces
    as DSDTM_ERR_HIP there — once. The other context's launch, in flight on its own stream meanwhile, and its check are
    unaffected (round 3 had one device-global flag: the first check of EITHER context reported and cleared it)."""
    gpu_ctx = gpu_ctx_diag            # the diagnostic library: this test needs its dsdtm_debug_* entries
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N = 320, 240, 3, 600
    other = capi.Context(0, diag=True)
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1900 + i, margin=12) for i in range(2)]
    ta, ba = _device_batch(torch, dev, scenes, L, W, Hh)
    tb, bb = _device_batch(torch, dev, scenes, L, W, Hh)
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    f, _ = _short_team(gpu_ctx)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    setopt = gpu_ctx.lib.dsdtm_debug_set_option
    try:
        gpu_ctx.check(setopt(b"no_recover", 1))
        gpu_ctx.check(f(gpu_ctx.handle, C.byref(ba), C.byref(cam), C.byref(prm), sa.cuda_stream))
        other.check(other.lib.dsdtm_sparse_align_batch_device(other.handle, C.byref(bb), C.byref(cam), C.byref(prm), sb.cuda_stream))
        assert other.lib.dsdtm_sparse_align_check(other.handle, sb.cuda_stream) == capi.OK
        rc = gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, sa.cuda_stream)
        assert rc == capi.ERR_HIP and b"timed out" in gpu_ctx.lib.dsdtm_last_error(gpu_ctx.handle)
        assert gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, sa.cuda_stream) == capi.OK     # reported once
        assert other.lib.dsdtm_sparse_align_check(other.handle, sb.cuda_stream) == capi.OK
    finally:
        gpu_ctx.check(setopt(b"no_recover", 0))
    Tg, ntg = tb["Tc"].cpu().numpy(), tb["nt"].cpu().numpy()
    for i, sc in enumerate(scenes):
        To, no, _ = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"the other context's pair {i}")
        assert ntg[i] == no


@pytest.mark.diag
def test_single_pair_run_recovers_from_a_team_timeout(gpu_ctx_diag, oracle):
    """Sprase_ImgAlign::Run through the resident-frame entry with 600 features runs as a team of three compute units. With
    the debug switch that keeps the last member away the first attempt times out; the entry re-seeds the pose from its
    pinned block and re-runs on one compute unit: the caller sees the oracle's result and no error."""
    gpu_ctx = gpu_ctx_diag            # the diagnostic library: this test needs its dsdtm_debug_* entries
    import ctypes as C
    sc = cached_scene(width=320, height=240, levels=3, n_patches=600, seed=1777, margin=12)
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    _, recovered = _short_team(gpu_ctx)
    drop = gpu_ctx.lib.dsdtm_debug_drop_team_members
    drop.restype = None
    drop.argtypes = [C.c_int]
    n0 = recovered(gpu_ctx.handle)
    try:
        drop(1)
        Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=gpu_ctx)
    finally:
        drop(0)
    assert recovered(gpu_ctx.handle) == n0 + 1
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what="single-pair re-run")
    assert ng == no and sg["iters"] == so["iters"]


@pytest.mark.diag
def test_team_exchange_survives_the_wrap_of_its_tag_epoch(oracle):
    """The team kernel's exchange words are tagged with 20 bits of launch epoch and never cleared between launches: ring slot
    and epoch repeat together every 2^20 team launches, and a word member m of pair p wrote exactly then would be accepted
    as this launch's partial. The library re-zeroes the ring when the epoch wraps. Here the counter is moved next to the wrap
    (dsdtm_debug_team_seq): teams of mixed sizes (5 members, 2, 5 again, other scenes) run on either side of it — after a
    launch with the SAME epoch and ring slot as the one behind the wrap left its words in the buffer — and every launch
    gives the oracle's results (Run is stateless across calls, src/Sprase_ImageAlign.cpp:22-27)."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    ctx = capi.Context(0, diag=True)                         # a context of its own: the counter is per context
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    seq = ctx.lib.dsdtm_debug_team_seq
    seq.restype, seq.argtypes = C.c_longlong, [C.c_void_p, C.c_longlong]
    cam = None
    prm = capi.AlignParams(L, 0, 10, 15)
    big = [cached_scene(width=W, height=Hh, levels=L, n_patches=1000, seed=900 + i, margin=12) for i in range(2)]
    big2 = [cached_scene(width=W, height=Hh, levels=L, n_patches=1000, seed=902 + i, margin=12) for i in range(2)][::-1]
    small = [cached_scene(width=W, height=Hh, levels=L, n_patches=500, seed=1900 + i, margin=12) for i in range(2)]
    cam = capi.camera_struct(big[0].cam)

    def run(scenes):
        t, b = _device_batch(torch, dev, scenes, L, W, Hh)
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), None))
        ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, None))
        Tg, ng = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
        for i, sc in enumerate(scenes):
            To, no, _ = oracle.sparse_align(sc, L, 0, 10)
            H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"{len(sc.px)} features, pair {i}")
            assert ng[i] == no

    assert seq(ctx.handle, 0) == 0
    run(big)                                                 # counter 1: epoch 1, ring slot 1 — its words stay behind
    seq(ctx.handle, 0xffffd)
    run(big)                                                 # 0xffffe
    run(small)                                               # 0xfffff: smaller teams, fewer words rewritten
    assert seq(ctx.handle, -1) == 0
    run(big2)                                                # wraps: epoch 1, ring slot 1 again, other scenes in the same words
    assert seq(ctx.handle, -1) == 1                          # the wrap was seen (and the ring re-zeroed)
    run(small)
    run(big)
    ctx.close()


@pytest.mark.parametrize("N,P,size", [(1000, 9, (320, 240)), (2000, 10, (640, 480)), (1500, 3, (320, 240)), (2048, 4, (640, 480))])
def test_workspace_kernels_walk_features_in_row_order_with_the_same_results(gpu_ctx_each, oracle, N, P, size):
    """Batches of large pairs order a pair's features by image row on the device (one LDS sort per pair) before the lanes take
    them: the reference sums in list order (src/Sprase_ImageAlign.cpp:84-103), so only the rounding of the sums may differ —
    the oracle's poses to 1e-8, identical n_tracked / iterations / exit codes / n_ref / n_vis, with the ordering on and off
    (debug option ws_no_sort), for ragged feature counts, uninitialised features, and feature pixels that are NaN, negative or
    far outside the image (the sort key is a hint: it must never become an index)."""
    gpu_ctx = gpu_ctx_each            # the release library, then the diagnostic one (its switches only there)
    import ctypes as C
    import copy
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh = size
    L = 3
    base = []
    for i in range(3):
        sc = copy.deepcopy(cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=5200 + N + i, margin=12))
        sc.initial[7::13] = 0
        sc.px[3] = (np.nan, 40.0); sc.px[4] = (-1e9, 1e9); sc.px[5] = (W + 500.0, -3.0); sc.px[6] = (np.inf, np.nan)
        sc.initial[3:7] = 0                                      # (the reference never reads the pixel of an uninitialised feature, :86)
        if N == 2048:                                            # the LAST index of a full list with a pixel that clamps to the last row
            sc.px[N - 1] = (1e9, 1e9); sc.initial[N - 1] = 0     # and column bin: its sort key must not collide with "no feature"
        base.append(sc)
    scenes = [base[i % 3] for i in range(P)]
    nf = np.array([N - (37 * i) % 300 for i in range(P)], np.int32)
    nf[0] = N
    t, b = _device_batch(torch, dev, scenes, L, W, Hh)
    t["nf"] = torch.from_numpy(nf).to(dev)
    b.n_features = t["nf"].data_ptr()
    t["st"] = torch.zeros((P, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    b.stats = t["st"].data_ptr()
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    seeds = torch.from_numpy(np.stack([s_.T_cur_w_seed.reshape(12) for s_ in scenes])).to(dev)
    outs = []
    import contextlib
    for no_sort in ((0, 1) if gpu_ctx.diag else (0,)):
        with (capi.debug_options(ws_no_sort=no_sort) if gpu_ctx.diag else contextlib.nullcontext()):
            t["Tc"].copy_(seeds)
            torch.cuda.synchronize()
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), None))
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, None))
        outs.append((t["Tc"].cpu().numpy().copy(), t["nt"].cpu().numpy().copy(),
                     np.frombuffer(t["st"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE).copy()))
    for i, sc in enumerate(scenes):
        sub = type("S", (), {})()
        for k in ("cam", "ref_pyr", "cur_pyr", "T_ref_w", "T_cur_w_seed"):
            setattr(sub, k, getattr(sc, k))
        sub.px, sub.bearing, sub.p_world, sub.initial = sc.px[:nf[i]], sc.bearing[:nf[i]], sc.p_world[:nf[i]], sc.initial[:nf[i]]
        To, no, so = oracle.sparse_align(sub, L, 0, 10)
        for (Tg, ng, sg), what in zip(outs, ("row order", "list order")):
            H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"{what}, pair {i}")
            assert ng[i] == no and list(sg["iters"][i][:L]) == list(so["iters"][:L]) and list(sg["exit_code"][i][:L]) == list(so["exit_code"][:L])
            assert list(sg["n_ref"][i][:L]) == list(so["n_ref"][:L]) and list(sg["n_vis"][i][:L]) == list(so["n_vis"][:L])
    if gpu_ctx.diag:
        assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-9         # the two orders differ by rounding only
    # and the ordered launch is deterministic: the same bits again
    t["Tc"].copy_(seeds)
    torch.cuda.synchronize()
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), None))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, None))
    assert np.array_equal(t["Tc"].cpu().numpy(), outs[0][0])


def test_workspace_launches_in_flight_on_two_streams(gpu_ctx, oracle):
    """Many pairs of more than 704 features run the workspace kernel, whose scratch belongs to the launch's
    STREAM: two such launches in flight on two streams of one context do not share it (round 1 had one
    workspace per context and forbade this)."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N, P = 320, 240, 3, 720, 70
    base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1500 + i, margin=12) for i in range(4)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
    groups = [[base[(i + g) % 4] for i in range(P)] for g in range(2)]
    packed = [_device_batch(torch, dev, g, L, W, Hh) for g in groups]
    cam = capi.camera_struct(base[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    streams = [torch.cuda.Stream(device=dev) for _ in groups]
    torch.cuda.synchronize()
    for rep in range(2):
        for (t, b), scs in zip(packed, groups):
            t["Tc"].copy_(torch.from_numpy(np.stack([s_.T_cur_w_seed.reshape(12) for s_ in scs])).to(dev))
        torch.cuda.synchronize()
        for (t, b), st in zip(packed, streams):
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
        for st in streams:
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, st.cuda_stream))
        for g, (t, b) in enumerate(packed):
            Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
            for i in range(P):
                To, no, _ = want[(i + g) % 4]
                H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"rep {rep} stream {g} pair {i}")
                assert ntg[i] == no


@pytest.mark.parametrize("N,scale", [(1000, 1.0), (1100, 2.5), (2040, 1.0), (2100, 1.0)])
def test_workspace_kernel_window_shapes(gpu_ctx, oracle, N, scale):
    """The workspace kernel keeps the current-image windows of up to 1024 / 2048 patches in dynamic LDS (64 / 128 KB)
    and gathers from the pyramid beyond that: every shape against the oracle, one of them with multi-pixel motion
    (windows are refilled when a patch leaves them)."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, P = 320, 240, 3, 66
    kw = dict(width=W, height=Hh, levels=L, n_patches=N, margin=12)
    if scale != 1.0:
        kw["xi"] = tuple(scale * v for v in (0.01, -0.006, 0.004, 0.004, -0.003, 0.005))
    base = [cached_scene(seed=1700 + i, **kw) for i in range(3)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
    scs = [base[i % 3] for i in range(P)]
    t, b = _device_batch(torch, dev, scs, L, W, Hh)
    cam = capi.camera_struct(base[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, st.cuda_stream))
    Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
    for i in range(P):
        To, no, _ = want[i % 3]
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"N {N} pair {i}")
        assert ntg[i] == no


def test_more_streams_than_a_context_tracks(gpu_ctx, oracle):
    """A context keeps pair counters (and workspaces) per stream for 16 streams; applications that keep creating
    streams get the least recently used entry handed over (one device synchronisation) — launches on 40 different
    streams, two in flight at any time, all give the oracle's results."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L = 320, 240, 3
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=150, seed=1700 + i, margin=12) for i in range(9)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in scenes]
    packed = [_device_batch(torch, dev, scenes, L, W, Hh) for _ in range(2)]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    seed = torch.from_numpy(np.stack([s_.T_cur_w_seed.reshape(12) for s_ in scenes])).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(40)]
    for k in range(0, 40, 2):
        for (t, b), st in zip(packed, streams[k:k + 2]):
            with torch.cuda.stream(st):
                t["Tc"].copy_(seed, non_blocking=True)
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
        for (t, b), st in zip(packed, streams[k:k + 2]):
            gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, st.cuda_stream))
            Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
            for i, (To, no, _) in enumerate(want):
                H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"stream {k} pair {i}")
                assert ntg[i] == no


def test_two_member_pairs_ragged_counts_determinism_and_the_one_unit_path(gpu_ctx_each, oracle):
    """1025..2048 patches in batches: one pair on TWO compute units (halves wholly in LDS, partials exchanged through
    tagged words). A pair count that is not a multiple of 8, ragged live counts — a member whose half holds no live
    patch, a pair below Min_fts — statistics from member 0, bit-identical results from launch to launch, and the same
    results (to rounding: other summation order) on one compute unit with the HBM workspace (option ws_no_duo)."""
    gpu_ctx = gpu_ctx_each            # the release library, then the diagnostic one (its switches only there)
    import ctypes as C
    import copy
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N, P = 320, 240, 3, 2000, 19
    base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=2300 + i, margin=12) for i in range(3)]
    scs = [base[i % 3] for i in range(P)]
    nf = np.array([2000, 1500, 1100, 900, 10] + [2000 - 37 * i for i in range(P - 5)], np.int32)
    want = []
    for i in range(P):
        s2 = copy.copy(scs[i]); n = int(nf[i])
        s2.px, s2.bearing, s2.p_world, s2.initial = s2.px[:n], s2.bearing[:n], s2.p_world[:n], s2.initial[:n]
        want.append(oracle.sparse_align(s2, L, 0, 10))
    t, b = _device_batch(torch, dev, scs, L, W, Hh)
    t["nf"] = torch.from_numpy(nf).to(dev)
    t["st"] = torch.zeros((P, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    b.n_features, b.stats = t["nf"].data_ptr(), t["st"].data_ptr()
    cam = capi.camera_struct(base[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    seed = torch.from_numpy(np.stack([s.T_cur_w_seed.reshape(12) for s in scs])).to(dev)

    def run():
        t["Tc"].copy_(seed); t["nt"].zero_(); t["st"].zero_()
        torch.cuda.synchronize()
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), None))
        gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, None))
        st = np.frombuffer(t["st"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
        return t["Tc"].cpu().numpy().copy(), t["nt"].cpu().numpy().copy(), st.copy()

    T1, n1, s1 = run()
    for i, (To, no, so) in enumerate(want):
        assert n1[i] == no, i
        if nf[i] < 15:
            assert np.array_equal(T1[i], scs[i].T_cur_w_seed.reshape(12))      # Min_fts rule: pose untouched
            continue
        H.assert_pose_close(T1[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"two members, pair {i}")
        assert list(s1["iters"][i][:L]) == list(so["iters"][:L]) and list(s1["exit_code"][i][:L]) == list(so["exit_code"][:L])
    for _ in range(3):
        T2, n2, s2 = run()
        assert np.array_equal(T1, T2) and np.array_equal(n1, n2) and np.array_equal(s1["chi2"], s2["chi2"])
    if gpu_ctx.diag:
        with capi.debug_options(ws_no_duo=1):
            T3, n3, s3 = run()
        assert np.array_equal(n1, n3) and np.array_equal(s1["iters"], s3["iters"]) and np.abs(T1 - T3).max() < 1e-12


def _roll_scene(theta, radius, n, seed, width=640, height=480, levels=4):
    """Features inside a disc around the principal point and a roll of `theta` rad between the frames: small pixel
    motion for a large angle, so the first Gauss-Newton step at the coarsest level is a rotation of ~theta."""
    import dataclasses
    sc = synth.make_scene(width=width, height=height, levels=levels, n_patches=n, seed=seed, xi=(0, 0, 0, 0, 0, theta), depth=2.0)
    rng = np.random.default_rng(seed)
    ang, r = rng.uniform(0, 2 * np.pi, n), radius * np.sqrt(rng.uniform(0, 1, n))
    px = np.stack([sc.cam.cx + r * np.cos(ang), sc.cam.cy + r * np.sin(ang)], axis=1).astype(np.float32)
    bearing = synth.bearing_from_px(sc.cam, px)
    return dataclasses.replace(sc, px=px, bearing=bearing, p_world=bearing * (sc.depth / bearing[:, 2:3]))


@pytest.mark.parametrize("n,batch", [(60, 0), (300, 0), (1000, 0), (800, 66), (1500, 66)])
def test_rotation_steps_beyond_a_tenth_of_a_radian(gpu_ctx, oracle, n, batch):
    """SE3::exp of a step with |omega|^2 >= 0.01 leaves the power series for the closed forms (sincos), which the
    kernels keep out of line: a roll of 0.15 / 0.25 rad seen by features near the principal point makes the first step
    at the coarsest level that large (checked on the oracle). One pair on the register kernels (60, 300 features) and on
    a team (1000); batches of 66 pairs (more teams than fit at once) on the workspace kernel (800) and on two compute
    units per pair (1500)."""
    import ctypes as C
    cases = [(0.15, 80.0), (0.25, 80.0)]
    scs = [_roll_scene(th, rad, n, seed=31 + k) for k, (th, rad) in enumerate(cases)]
    for sc, (th, rad) in zip(scs, cases):
        T1, _, _ = oracle.sparse_align(sc, 4, 3, 1)                  # the first step alone
        assert synth.pose_error(T1, sc.T_cur_w_seed)[0] > 0.1, "the scene does not produce a large step"
    want = [oracle.sparse_align(sc, 4, 0, 10) for sc in scs]
    if batch == 0:
        for sc, (To, no, so) in zip(scs, want):
            Tg, ng, sg = H.gpu_sparse_align(sc, 4, 0, 10, ctx=gpu_ctx)
            H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"n {n}")
            assert ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"]
            assert synth.pose_error(Tg, sc.T_cur_w_true)[0] < 0.2 * abs(np.arccos((np.trace(sc.T_cur_w_true[:, :3]) - 1) / 2))   # and it converges
        return
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    group = [scs[i % 2] for i in range(batch)]
    t, b = _device_batch(torch, dev, group, 4, 640, 480)
    t["st"] = torch.zeros((batch, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    b.stats = t["st"].data_ptr()
    cam = capi.camera_struct(scs[0].cam)
    prm = capi.AlignParams(4, 0, 10, 15)
    torch.cuda.synchronize()
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_batch_device(gpu_ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), None))
    gpu_ctx.check(gpu_ctx.lib.dsdtm_sparse_align_check(gpu_ctx.handle, None))
    Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
    st = np.frombuffer(t["st"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
    for i in range(batch):
        To, no, so = want[i % 2]
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"n {n} pair {i}")
        assert ntg[i] == no and list(st["iters"][i][:4]) == list(so["iters"][:4]) and list(st["exit_code"][i][:4]) == list(so["exit_code"][:4])


@pytest.mark.parametrize("N", [800, 1500, 2000])
def test_workspace_launches_are_graph_capturable_after_reserve(gpu_ctx, oracle, N):
    """Batches of large pairs inside a hipGraph: up to 1024 patches everything is in LDS (nothing to reserve); beyond,
    a live launch grows its stream's workspace on demand, which a captured launch cannot — it is refused with a message
    until dsdtm_reserve(dsdtm_sparse_align_workspace_bytes(batch)) has run, and then uses the context's reserved
    workspace: 1500 patches park their grid inputs there, 2000 run on two compute units per pair and keep their
    exchange words there (zeroed by a memset node of the graph at every replay)."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, P = 320, 240, 3, 66
    base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=2500 + i, margin=12) for i in range(3)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
    scs = [base[i % 3] for i in range(P)]
    t, b = _device_batch(torch, dev, scs, L, W, Hh)
    seed = t["Tc"].clone()
    cam = capi.camera_struct(base[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    ctx = capi.Context(0)                                        # a context of its own: nothing reserved yet
    need = ctx.lib.dsdtm_sparse_align_workspace_bytes(C.byref(b))
    assert (need == 0) == (N <= 1024)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    if need:
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=side):
            t["Tc"].copy_(seed)
            rc = ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm),
                                                         torch.cuda.current_stream().cuda_stream)
        assert rc == capi.ERR_INVALID and b"dsdtm_reserve" in ctx.lib.dsdtm_last_error(ctx.handle)
        del g0
        ctx.check(ctx.lib.dsdtm_reserve(ctx.handle, need))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        t["Tc"].copy_(seed)
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm),
                                                          torch.cuda.current_stream().cuda_stream))
    first = None
    for rep in range(3):
        t["Tc"].zero_(); t["nt"].zero_()
        g.replay()
        ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, None))
        torch.cuda.synchronize()
        Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
        for i in range(P):
            To, no, _ = want[i % 3]
            H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"N {N} replay {rep} pair {i}")
            assert ntg[i] == no
        if first is None:
            first = Tg.copy()
        assert np.array_equal(first, Tg)
    del g
    ctx.close()


@pytest.mark.diag
def test_more_unsettled_multi_cu_launches_than_recovery_slots(gpu_ctx_diag, oracle):
    """A context keeps what a re-run needs for 64 unsettled multi-CU launches. The first of 70 team launches on one stream
    loses a member (its waits run out); nobody calls dsdtm_sparse_align_check in between, so the 65th launch has to settle
    the oldest one itself — waits for its event, finds the timeout word set, re-seeds and re-runs it on the one-CU kernels —
    before it reuses the slot. Every launch ends with the oracle's results and the final check has nothing left to repair."""
    gpu_ctx = gpu_ctx_diag            # the diagnostic library: this test needs its dsdtm_debug_* entries
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N = 320, 240, 3, 600
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1300 + N + i, margin=12) for i in range(2)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in scenes]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    ctx = capi.Context(0, diag=True)                       # its own context: all 64 slots free
    f, recovered = _short_team(ctx)
    st = torch.cuda.Stream(device=dev)
    packed = [_device_batch(torch, dev, scenes, L, W, Hh) for _ in range(70)]
    for k, (t, b) in enumerate(packed):
        launch = f if k == 0 else ctx.lib.dsdtm_sparse_align_batch_device
        ctx.check(launch(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
        assert recovered(ctx.handle) == (1 if k >= 64 else 0)       # launch 64 (the 65th) settled launch 0
    ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, st.cuda_stream))
    assert recovered(ctx.handle) == 1
    for k, (t, b) in enumerate(packed):
        Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
        for i, (To, no, _) in enumerate(want):
            H.assert_pose_close(Tg[i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"launch {k} pair {i}")
            assert ntg[i] == no
    ctx.close()


@pytest.mark.diag
def test_captured_large_pairs_run_the_one_cu_kernels(gpu_ctx_diag, oracle):
    """Inside a stream capture no multi-CU kernel is used (a graph replay could be neither ordered against live team
    launches nor re-run after a timeout): 18 pairs of 1900 features captured into a hipGraph run the one-CU workspace kernel
    (scratch reserved up front), replay twice with the oracle's results, and leave nothing to settle."""
    gpu_ctx = gpu_ctx_diag            # the diagnostic library: this test needs its dsdtm_debug_* entries
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N, P = 320, 240, 3, 1900, 18
    base = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1300 + N + i, margin=12) for i in range(3)]
    scenes = [base[i % 3] for i in range(P)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in base]
    t, b = _device_batch(torch, dev, scenes, L, W, Hh)
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    ctx = capi.Context(0, diag=True)
    _, recovered = _short_team(ctx)
    ctx.check(ctx.lib.dsdtm_reserve(ctx.handle, ctx.lib.dsdtm_sparse_align_workspace_bytes(C.byref(b))))
    seed = t["Tc"].clone()
    st = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        t["Tc"].copy_(seed)
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), torch.cuda.current_stream().cuda_stream))
    for rep in range(2):
        t["Tc"].zero_()
        g.replay()
        torch.cuda.synchronize()
        Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
        for i in range(P):
            To, no, _ = want[i % 3]
            H.assert_pose_close(Tg[i].reshape(3, 4), To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"replay {rep} pair {i}")
            assert ntg[i] == no
    ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, st.cuda_stream))
    assert recovered(ctx.handle) == 0
    ctx.close()


@pytest.mark.diag
def test_a_timed_out_launch_whose_pose_buffer_a_later_launch_reuses_is_reported(oracle):
    """A re-run after a timeout is queued BEHIND the stream's later work. Two unchecked multi-CU launches of one stream into the
    SAME pose buffer, the first of which times out (a member stays away): re-running it would overwrite the second launch's
    results with the first's, so the check reports DSDTM_ERR_HIP instead of returning OK with the wrong poses (round-5 advice;
    dsdtm_amd.h: unchecked launches of one stream name distinct output buffers). The context stays usable."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    dev = torch.device("cuda", 0)
    W, Hh, L, N = 320, 240, 3, 600
    scenes = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1300 + N + i, margin=12) for i in range(2)]
    want = [oracle.sparse_align(sc, L, 0, 10) for sc in scenes]
    cam = capi.camera_struct(scenes[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    ctx = capi.Context(0, diag=True)
    f, recovered = _short_team(ctx)
    st = torch.cuda.Stream(device=dev)
    t, b = _device_batch(torch, dev, scenes, L, W, Hh)
    ctx.check(f(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))                                   # times out
    ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))   # same buffers, no check between
    rc = ctx.lib.dsdtm_sparse_align_check(ctx.handle, st.cuda_stream)
    assert rc == capi.ERR_HIP and b"reuses" in ctx.lib.dsdtm_last_error(ctx.handle)
    assert recovered(ctx.handle) == 0                                           # nothing was re-run over the later launch's results
    assert ctx.lib.dsdtm_sparse_align_check(ctx.handle, st.cuda_stream) == capi.OK      # reported once; the second record settles clean
    # the context is usable: a fresh launch from the seeds gives the oracle's results
    t["Tc"].copy_(torch.from_numpy(np.stack([s.T_cur_w_seed.reshape(12) for s in scenes])).to(dev))
    torch.cuda.synchronize()
    ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cam), C.byref(prm), st.cuda_stream))
    ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, st.cuda_stream))
    Tg, ntg = t["Tc"].cpu().numpy(), t["nt"].cpu().numpy()
    for i, (To, no, _) in enumerate(want):
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"after the refused re-run, pair {i}")
        assert ntg[i] == no
    ctx.close()
