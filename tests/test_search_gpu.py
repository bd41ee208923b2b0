"""SearchLocalPoints: speculative GPU matching + host replay vs the sequential CPU restatement."""
import copy

import numpy as np
import pytest

from dsdtm_amd import search, synth
from dsdtm_amd.frame import Config, Frame
from tests import search_restatement as SR


def make_world(seed, n_points=900, n_kf=3, width=640, height=480, cell=25, obs_margin=12, uv_margin=20, tex=None, cam=None, uv=None):
    """A textured plane seen by n_kf keyframes and one current frame; map points on the plane with
    observations in the keyframes. `tex` (a real image), `cam` and `uv` (where on the first keyframe's image the map points sit,
    e.g. a detector's corners) replace the seeded texture, the TUM intrinsics and the uniform draw."""
    rng = np.random.default_rng(seed)
    cam = synth.Camera.tum(width, height) if cam is None else cam
    tex = synth.make_texture(height, width, seed) if tex is None else np.asarray(tex, np.float64)
    depth = 2.0
    frames = []
    poses = [np.eye(4)] + [synth.se3_exp(np.concatenate([rng.uniform(-0.06, 0.06, 3), rng.uniform(-0.03, 0.03, 3)])) for _ in range(n_kf)]
    imgs = [np.clip(np.rint(tex), 0, 255).astype(np.uint8)] + [synth.warp_plane(tex, cam, T, depth) for T in poses[1:]]
    kfs = [search.KeyFrame(cam, synth.build_pyramid(imgs[i], 5), poses[i][:3], i) for i in range(n_kf)]
    cur = Frame(cam, synth.build_pyramid(imgs[n_kf], 5), poses[n_kf][:3])
    # map points: plane points (world == first keyframe's camera frame)
    if uv is None:
        uv = np.stack([rng.uniform(uv_margin, width - uv_margin, n_points), rng.uniform(uv_margin, height - uv_margin, n_points)], 1)
    else:
        uv = np.asarray(uv, np.float64); n_points = len(uv)
    ray = np.stack([(uv[:, 0] - cam.cx) / cam.fx, (uv[:, 1] - cam.cy) / cam.fy, np.ones(n_points)], 1)
    P = ray * depth
    feats = [[] for _ in range(n_kf)]
    mps = []
    for i in range(n_points):
        obs = {}
        for k in range(n_kf):
            if rng.random() < 0.7:
                px = kfs[k].World2Pixel(P[i])
                lvl = int(rng.integers(0, 2))
                if obs_margin * (1 << lvl) < px[0] < width - obs_margin * (1 << lvl) and obs_margin * (1 << lvl) < px[1] < height - obs_margin * (1 << lvl):
                    obs[k] = len(feats[k])
                    feats[k].append((px.astype(np.float32), lvl))
        mps.append(search.MapPoint(P[i].copy(), obs, mnFound=int(rng.integers(1, 6)), mbBad=bool(rng.random() < 0.03)))
    for k in range(n_kf):
        px = np.array([f[0] for f in feats[k]], np.float32).reshape(-1, 2)
        kfs[k].set_features(px, synth.bearing_from_px(cam, px), np.zeros((len(px), 3)), np.ones(len(px), np.uint8),
                            level=np.array([f[1] for f in feats[k]], np.int32))
    return cam, kfs, cur, mps


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_search_local_points_matches_sequential_reference_flow(gpu_ctx, seed):
    Config.Set("Camera.CellSize", 25)
    Config.Set("Camera.MaxPyraLevels", 5)
    cam, kfs, cur, mps = make_world(seed)
    s = search.LocalPointSearch(cam, ctx=gpu_ctx)
    s.ResetGrid()
    n_in = sum(s.ReprojectPoint(cur, mp) for mp in mps)
    assert n_in > 500
    cells_copy = [[[c[0], c[1].copy()] for c in cell] for cell in s.mCells]
    mps_before = {id(mp): mp.mnFound for mp in mps}
    mask_g = np.full((cam.height, cam.width), 255, np.uint8)
    got = s.SearchLocalPoints(cur, kfs, mask_g)
    # sequential restatement on an identical copy of the state
    for mp in mps:
        mp.mnFound = mps_before[id(mp)]
    mask_o = np.full((cam.height, cam.width), 255, np.uint8)
    want = SR.search_local_points(cells_copy, cur, kfs, cam, 25, 5, mask_o)
    assert len(got) == len(want) and len(got) >= 150, (len(got), len(want))
    assert [g[0] for g in got] == [w[0] for w in want]                    # same cells, same order
    assert all(g[1] is w[1] for g, w in zip(got, want))                   # same map point per cell
    assert [g[3] for g in got] == [w[3] for w in want]                    # same search level
    assert np.array_equal(np.array([g[2] for g in got]), np.array([w[2] for w in want]))   # same refined pixels, bit for bit
    assert np.array_equal(mask_g, mask_o)
    assert len(got) <= 200


@pytest.mark.gpu
def test_search_on_resident_frames_is_identical(gpu_ctx):
    """dsdtm_match_candidates_frames (warp prelude + Align2D in one call on device-resident frames) gives
    the two-call host path's matches bit for bit."""
    Config.Set("Camera.CellSize", 25)
    Config.Set("Camera.MaxPyraLevels", 5)
    outs = []
    for resident in (False, True):
        cam, kfs, cur, mps = make_world(7, n_points=600)
        s = search.LocalPointSearch(cam, ctx=gpu_ctx, resident_frames=resident)
        s.ResetGrid()
        for mp in mps:
            s.ReprojectPoint(cur, mp)
        mask = np.full((cam.height, cam.width), 255, np.uint8)
        got = s.SearchLocalPoints(cur, kfs, mask)
        idx = {id(mp): i for i, mp in enumerate(mps)}
        outs.append(([(g[0], idx[id(g[1])], float(g[2][0]), float(g[2][1]), g[3]) for g in got], mask))
    assert outs[0][0] == outs[1][0] and len(outs[0][0]) > 100
    assert np.array_equal(outs[0][1], outs[1][1])


def test_fill_circle_matches_independent_version():
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = np.full((60, 80), 255, np.uint8); b = a.copy()
        cx, cy, r = int(rng.integers(-5, 85)), int(rng.integers(-5, 65)), int(rng.integers(0, 30))
        search.fill_circle(a, cx, cy, r, 0)
        SR.circle_filled(b, cx, cy, r)
        assert np.array_equal(a, b)
        yy, xx = np.mgrid[0:60, 0:80]
        disc = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        assert (a[disc & ((xx - cx) ** 2 + (yy - cy) ** 2 <= (r - 1) ** 2 if r > 0 else disc)] == 0).all()   # interior is painted


def test_grid_and_reproject_point():
    Config.Set("Camera.CellSize", 25)
    cam = synth.Camera.tum(640, 480)
    s = search.LocalPointSearch.__new__(search.LocalPointSearch)
    search.FA.Feature_Alignment.__init__(s, cam, None)
    s.mCell_size = 25
    s.mGrid_Rows, s.mGrid_Cols = int(np.ceil(480 / 25)), int(np.ceil(640 / 25))
    s.mCells = [[] for _ in range(s.mGrid_Rows * s.mGrid_Cols)]
    assert (s.mGrid_Rows, s.mGrid_Cols) == (20, 26)
    f = Frame(cam, [np.zeros((480, 640), np.uint8)], np.eye(4)[:3])
    mp_in = search.MapPoint(np.array([0.0, 0.0, 2.0]), {})
    mp_out = search.MapPoint(np.array([5.0, 0.0, 2.0]), {})
    assert s.ReprojectPoint(f, mp_in) and not s.ReprojectPoint(f, mp_out)
    px = f.World2Pixel(mp_in.mPose)
    assert len(s.mCells[int(px[1] / 25) * 26 + int(px[0] / 25)]) == 1


@pytest.mark.gpu
def test_match_candidates_batch_device_equals_per_frame_calls(gpu_ctx):
    """dsdtm_match_candidates_batch_device: the candidates of THREE current frames (three independent worlds, nine
    keyframes) in one call on packed device pyramids — convergence flags, refined pixels and search levels bit for bit
    those of dsdtm_match_candidates_frames called once per current frame."""
    import ctypes as C
    import torch
    from dsdtm_amd import capi
    from dsdtm_amd import feature_alignment as FA
    Config.Set("Camera.CellSize", 25)
    Config.Set("Camera.MaxPyraLevels", 5)
    dev = torch.device("cuda", 0)
    L, W, Hh = 5, 640, 480
    ws, hs, ss, offs, nbytes = capi.pyramid_layout(W, Hh, L)
    pitch = (nbytes + 255) // 256 * 256

    def pack(pyr):
        out = np.zeros(pitch, np.uint8)
        for l in range(L):
            out[offs[l]:offs[l] + ws[l] * hs[l]] = pyr[l].reshape(-1)
        return out

    per_frame, cols = [], {k: [] for k in ("frame", "kf", "rp", "rl", "rb", "pw", "px")}
    cur_pack, kf_pack, Tk_all, Tc_all = [], [], [], []
    for wi, seed in enumerate((11, 12, 13)):
        cam, kfs, cur, mps = make_world(seed, n_points=300)
        ck, rp, rl, rb, pw, cpx = [], [], [], [], [], []
        for mp in mps:
            if mp.mbBad or not mp.mObservations:
                continue
            px = cur.World2Pixel(mp.Get_Pose())
            if not (20 < px[0] < W - 20 and 20 < px[1] < Hh - 20):
                continue
            k, fi = sorted(mp.mObservations.items())[0]
            ck.append(k); rp.append(kfs[k].px[fi]); rl.append(kfs[k].level[fi]); rb.append(kfs[k].bearing[fi]); pw.append(mp.Get_Pose()); cpx.append(px)
        ck, rl = np.array(ck, np.int32), np.array(rl, np.int32)
        rp, rb, pw, cpx = np.array(rp, np.float32), np.array(rb), np.array(pw), np.array(cpx)
        Tk = np.array([k.Get_Pose() for k in kfs])
        per_frame.append(FA.match_candidates_frames(cur, kfs, cam, Tk, cur.Get_Pose(), ck, rp, rl, rb, pw, cpx, L - 3, 10, ctx=gpu_ctx))
        cols["frame"].append(np.full(len(ck), wi, np.int32)); cols["kf"].append(ck + 3 * wi)
        for k, v in (("rp", rp), ("rl", rl), ("rb", rb), ("pw", pw), ("px", cpx)):
            cols[k].append(v)
        cur_pack.append(pack(cur.mvImg_Pyr)); kf_pack += [pack(k.mvImg_Pyr) for k in kfs]
        Tk_all.append(Tk.reshape(3, 12)); Tc_all.append(np.asarray(cur.Get_Pose()).reshape(12))
    cat = {k: np.ascontiguousarray(np.concatenate(v)) for k, v in cols.items()}
    M = len(cat["frame"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_cur, d_kf = t(np.stack(cur_pack)), t(np.stack(kf_pack))
    d_Tk, d_Tc = t(np.concatenate(Tk_all)), t(np.stack(Tc_all))
    d = {k: t(v) for k, v in cat.items()}
    d_px = d["px"].clone()
    d_sl = torch.zeros(M, dtype=torch.int32, device=dev)
    d_cv = torch.zeros(M, dtype=torch.uint8, device=dev)
    d_scr = torch.empty(gpu_ctx.lib.dsdtm_match_candidates_scratch_bytes(M), dtype=torch.uint8, device=dev)
    wa, ha, sa, oa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*ss), (C.c_size_t * L)(*offs)
    gpu_ctx.check(gpu_ctx.lib.dsdtm_match_candidates_batch_device(
        gpu_ctx.handle, d_cur.data_ptr(), 3, d_kf.data_ptr(), 9, pitch, L, wa, ha, sa, oa, C.byref(capi.camera_struct(cam)),
        d_Tk.data_ptr(), d_Tc.data_ptr(), d["frame"].data_ptr(), d["kf"].data_ptr(), d["rp"].data_ptr(), d["rl"].data_ptr(),
        d["rb"].data_ptr(), d["pw"].data_ptr(), L - 3, 10, M, d_scr.data_ptr(), d_px.data_ptr(), d_sl.data_ptr(), d_cv.data_ptr(), None))
    torch.cuda.synchronize()
    conv, px, sl = d_cv.cpu().numpy().astype(bool), d_px.cpu().numpy(), d_sl.cpu().numpy()
    o = 0
    for cv_f, px_f, sl_f in per_frame:
        n = len(cv_f)
        assert np.array_equal(conv[o:o + n], cv_f) and np.array_equal(sl[o:o + n], sl_f)
        assert np.array_equal(px[o:o + n], px_f, equal_nan=True)
        o += n
    assert o == M and conv.sum() > 100
    # candidates that name a current frame outside the batch are rejected on the device, not dereferenced (the frame
    # index is data: a corrupted one must not become an out-of-bounds read of cur_pyr / T_cur_w)
    fr = d["frame"].clone()
    bad = torch.tensor([0, 5, M - 1], device=dev)
    fr[bad] = torch.tensor([-1, 3, 1 << 20], dtype=fr.dtype, device=dev)
    px_in = cat["px"]
    d_px2, d_sl2, d_cv2 = d["px"].clone(), torch.full_like(d_sl, 7), torch.ones_like(d_cv)
    gpu_ctx.check(gpu_ctx.lib.dsdtm_match_candidates_batch_device(
        gpu_ctx.handle, d_cur.data_ptr(), 3, d_kf.data_ptr(), 9, pitch, L, wa, ha, sa, oa, C.byref(capi.camera_struct(cam)),
        d_Tk.data_ptr(), d_Tc.data_ptr(), fr.data_ptr(), d["kf"].data_ptr(), d["rp"].data_ptr(), d["rl"].data_ptr(),
        d["rb"].data_ptr(), d["pw"].data_ptr(), L - 3, 10, M, d_scr.data_ptr(), d_px2.data_ptr(), d_sl2.data_ptr(), d_cv2.data_ptr(), None))
    torch.cuda.synchronize()
    conv2, px2, sl2 = d_cv2.cpu().numpy().astype(bool), d_px2.cpu().numpy(), d_sl2.cpu().numpy()
    badn = bad.cpu().numpy()
    assert not conv2[badn].any() and (sl2[badn] == -1).all() and np.array_equal(px2[badn], px_in[badn])
    keep = np.ones(M, bool); keep[badn] = False
    assert np.array_equal(conv2[keep], conv[keep]) and np.array_equal(sl2[keep], sl[keep]) and np.array_equal(px2[keep], px[keep], equal_nan=True)


@pytest.mark.gpu
def test_match_candidates_batch_device_equals_the_oracle_across_group_boundaries():
    """Candidate counts on both sides of the warp prelude's group sizes (2 and 16 per group, switched at 8192) and Align2D's
    16 features per group, with random invalid candidates mixed in: search level, flag and refined pixel equal the CPU
    oracle's warp + Align2D bit for bit (tools/soak_fmd.py runs the long list)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("soak_fmd", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_fmd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main((1, 17, 127, 8193)) == 0


@pytest.mark.gpu
@pytest.mark.diag
def test_find_match_direct_in_one_kernel_equals_the_two_kernel_path_and_the_oracle():
    """Round 5: FindMatchDirect is one launch (match.hip: the warped patches never leave LDS). The two-kernel path of rounds 1-4
    (warp_kernel -> HBM -> align2d_rows_kernel; debug option fmd_split) runs the same device functions: search level, flag and
    refined pixel of both equal the CPU oracle's warp + Align2D bit for bit, across the group boundaries of both shapes."""
    import importlib.util
    import os
    from dsdtm_amd import capi
    spec = importlib.util.spec_from_file_location("soak_fmd", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_fmd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main((1, 15, 16, 17, 127, 4099)) == 0                 # the release library: one kernel
    with capi.diag_default(), capi.debug_options(fmd_split=1):       # the diagnostic library: the two-kernel path of rounds 1-4
        assert mod.main((1, 15, 16, 17, 127, 4099)) == 0
