"""dsdtm_track_frame — one tracked frame (reference src/Tracking.cpp:199-256) in ONE submission: new frame -> Run ->
ReprojectPoint + Get_ClosetObs -> FindMatchDirect for every point -> the cell walk of SearchLocalPoints replayed ON THE DEVICE
-> PoseOptimization. Held to
  * the sequential CPU restatement of the search (tests/search_restatement.py) on the quirk worlds, and told apart from all eight
    S1 / W3 mutants of it at the case and check tests/mutant_runs.py names;
  * the four-call chain (Sprase_ImgAlign / LocalPointSearch / Optimizer on device-resident frames) bit for bit over a tracked
    sequence, map side effects included;
  * the C ABI's argument checks."""
import copy
import ctypes as C

import numpy as np
import pytest

from dsdtm_amd import capi, search, synth, tracking
from dsdtm_amd.frame import Config, Frame
from dsdtm_amd.optimizer import Optimizer
from dsdtm_amd.sparse_align import Sprase_ImgAlign
from tests import helpers as H
from tests import mutant_runs as M
from tests import quirk_fixtures as Q
from tests.test_search_gpu import make_world

pytestmark = pytest.mark.gpu


def track_search(ctx, name, **kw):
    """A named search world through dsdtm_track_frame with Run switched off (no reference features, min_tracked 0: the pose stays
    the seed = the world's current pose): (match list, mask painted from it) in the shape of quirk_fixtures.search_restated."""
    cam, kfs, cur, mps, cell = Q.search_world(name)
    last = Frame(cam, cur.mvImg_Pyr, cur.Get_Pose())            # any resident frame of the geometry; it has no features: Run returns 0
    r = tracking.track_frame(ctx, cam, cur.mvImg_Pyr[0], 5, last, cur.Get_Pose(), (5, 0, 8, 15), 0, kfs, mps, cell_size=cell,
                             max_pyr_levels=5, **kw)
    assert r["n_tracked"] == 0 and not r["lost"] and np.array_equal(r["T_run"], cur.Get_Pose())
    mask = np.full((cam.height, cam.width), 255, np.uint8)
    m = r["matches"]
    for q in m["px"]:
        search.fill_circle(mask, search.cvRound(float(q[0])), search.cvRound(float(q[1])), cell, 0)
    r["frame"].close()
    return [(int(m["cell"][k]), int(m["point"][k]), float(m["px"][k][0]), float(m["px"][k][1]), int(m["level"][k])) for k in range(len(m))], mask, r


@pytest.fixture(scope="module")
def track_outputs(gpu_ctx):
    return {name: track_search(gpu_ctx, name) for name in Q.SEARCH_WORLDS}


@pytest.mark.parametrize("name", list(Q.SEARCH_WORLDS))
def test_device_replay_equals_the_sequential_restatement(track_outputs, name):
    lst, mask, r = track_outputs[name]
    want = Q.search_restated(name)
    assert Q.search_first_difference(want, (lst, mask)) is None
    assert len(lst) >= 150 and len(lst) <= 200 and r["n_in_grid"] > 500
    # and the committed outputs of the four-call chain (tests/golden/quirks.npz)
    g = np.load(H.golden_path("quirks.npz"))
    assert np.array_equal(np.array(lst, np.float64).reshape(-1, 5), g[f"search_{name}_matches"])
    assert np.array_equal(np.packbits(mask == 255, axis=1), g[f"search_{name}_mask_rows"])


@pytest.mark.parametrize("mutant", [m for m in M.ALL_MUTANTS if M.domain(m) == "search"])
def test_device_replay_disagrees_with_every_search_mutant(track_outputs, mutant):
    case, check, quirk, cite = M.TABLE[mutant]
    name = case.split(":")[1]
    out = M.cpu_outputs(mutant, domains=("search",), search_worlds=[name])
    lst, mask, _ = track_outputs[name]
    got = Q.search_first_difference(out[case], (lst, mask))
    assert got == check, f"{mutant} ({quirk}, {cite}): {got!r} for the device replay against this mutant, expected {check!r}"


def test_search_part_equals_the_speculative_host_replay_and_honours_a_mask(gpu_ctx):
    """The same world through LocalPointSearch (speculative batch + HOST replay) and through the device replay, with a caller's
    mask that already blocks a band of the image (Frame::mImgMask at the start of the search): same matches, same final mask."""
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5)
    cam, kfs, cur, mps = make_world(9, n_points=800)
    mask0 = np.full((cam.height, cam.width), 255, np.uint8)
    mask0[150:260, :] = 0
    mask0[:, 400:470] = 7
    mps_h = copy.deepcopy(mps)
    s = search.LocalPointSearch(cam, ctx=gpu_ctx, resident_frames=True)
    s.ResetGrid()
    for mp in mps_h:
        if not mp.IsBad():
            s.ReprojectPoint(cur, mp)
    mask_h = mask0.copy()
    idx = {id(mp): i for i, mp in enumerate(mps_h)}
    want = [(g[0], idx[id(g[1])], float(g[2][0]), float(g[2][1]), g[3]) for g in s.SearchLocalPoints(cur, kfs, mask_h)]
    last = Frame(cam, cur.mvImg_Pyr, cur.Get_Pose())
    r = tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, cur.Get_Pose(), (5, 0, 8, 15), 0, kfs, mps, mask=mask0.copy())
    m = r["matches"]
    got = [(int(m["cell"][k]), int(m["point"][k]), float(m["px"][k][0]), float(m["px"][k][1]), int(m["level"][k])) for k in range(len(m))]
    assert got == want and 40 < len(got) < 200
    mask_d = mask0.copy()
    for q in m["px"]:
        search.fill_circle(mask_d, search.cvRound(float(q[0])), search.cvRound(float(q[1])), 25, 0)
    assert np.array_equal(mask_d, mask_h)
    r["frame"].close()


@pytest.mark.parametrize("n_points,cell", [(4000, 25), (4096, 10), (4096, 40)])
def test_large_and_dense_local_maps(gpu_ctx, n_points, cell):
    """Up to the documented 4096 map points (four ranks per thread of the replay workgroup), small cells, and cells so
    large that a candidate's neighbourhood holds more than 64 earlier ranks (the workgroup then scans every earlier candidate:
    replay_full_scan): the device replay gives the host replay's matches (LocalPointSearch: speculative batch + host walk)."""
    Config.Set("Camera.CellSize", cell); Config.Set("Camera.MaxPyraLevels", 5)
    cam, kfs, cur, mps = make_world(13, n_points=n_points, cell=cell, obs_margin=3)
    mps_h = copy.deepcopy(mps)
    s = search.LocalPointSearch(cam, ctx=gpu_ctx, resident_frames=True)
    s.ResetGrid()
    for mp in mps_h:
        if not mp.IsBad():
            s.ReprojectPoint(cur, mp)
    idx = {id(mp): i for i, mp in enumerate(mps_h)}
    want = [(g[0], idx[id(g[1])], float(g[2][0]), float(g[2][1]), g[3]) for g in s.SearchLocalPoints(cur, kfs)]
    last = Frame(cam, cur.mvImg_Pyr, cur.Get_Pose())
    r = tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, cur.Get_Pose(), (5, 0, 8, 15), 0, kfs, mps, cell_size=cell)
    m = r["matches"]
    got = [(int(m["cell"][k]), int(m["point"][k]), float(m["px"][k][0]), float(m["px"][k][1]), int(m["level"][k])) for k in range(len(m))]
    assert got == want and len(got) >= 100
    if cell == 40:
        assert r["replay_full_scan"]          # 4096 candidates over 192 cells: > 64 ranks in a five-cell row range
    if cell == 10:
        assert not r["replay_full_scan"]      # 1.3 candidates per cell: the rank-range masks
    r["frame"].close()
    Config.Set("Camera.CellSize", 25)


def test_tracked_sequence_one_call_per_frame_equals_the_four_call_chain(gpu_ctx):
    """Seven frames tracked twice on copies of one world: through the four synchronous calls per frame (Run, SearchLocalPoints'
    speculative batch + host replay, PoseOptimization — tests/test_tracking_sequence_gpu.py holds that chain to the CPU oracle)
    and through ONE dsdtm_track_frame per frame. Every pose, count, iteration, match, refined pixel, residual decision and
    map side effect (found counts, bad flags) must be the same, bit for bit, frame after frame — frame k is the reference
    frame of frame k + 1, so any difference would also compound."""
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    n_kf, n_frames = 2, 7
    cam, kfs, _, mps = make_world(21, n_points=700, n_kf=n_kf)
    rng = np.random.default_rng(77)
    tex = synth.make_texture(cam.height, cam.width, 21)
    for k, kf in enumerate(kfs):
        mpts = [None] * kf.n_features
        for mp in mps:
            if k in mp.mObservations:
                mpts[mp.mObservations[k]] = mp
        kf.mvMapPoints = mpts
        kf.p_world = np.array([m_.mPose if m_ is not None else np.zeros(3) for m_ in mpts])
        kf.initial = np.array([1 if m_ is not None else 0 for m_ in mpts], np.uint8)
    worlds = [copy.deepcopy((kfs, mps)) for _ in range(2)]
    T0 = np.vstack([kfs[n_kf - 1].Get_Pose(), [0, 0, 0, 1]])
    imgs, xi = [], np.zeros(6)
    for k in range(n_frames):
        xi = xi + np.concatenate([rng.uniform(-0.012, 0.012, 3), rng.uniform(-0.006, 0.006, 3)])
        imgs.append(synth.warp_plane(tex, cam, synth.se3_exp(xi) @ T0, 2.0))

    # ---- four calls per frame ----
    a_kfs, a_mps = worlds[0]
    al = Sprase_ImgAlign(5, 0, 8, ctx=gpu_ctx, resident_frames=True)
    srch = search.LocalPointSearch(cam, ctx=gpu_ctx, resident_frames=True)
    a_idx = {id(mp): i for i, mp in enumerate(a_mps)}
    a_log, last = [], a_kfs[n_kf - 1]
    for k in range(n_frames):
        cur = Frame(cam, synth.build_pyramid(imgs[k], 5), last.Get_Pose())
        n = al.Run(cur, last)
        T_run = cur.Get_Pose().copy()
        srch.ResetGrid()
        for mp in a_mps:
            if not mp.IsBad():
                srch.ReprojectPoint(cur, mp)
        matches = srch.SearchLocalPoints(cur, a_kfs)
        sm = Optimizer.PoseOptimization(cur, ctx=gpu_ctx)
        a_log.append(dict(n=n, T_run=T_run, iters=list(al.last_stats["iters"]), matches=[(m[0], a_idx[id(m[1])], m[3]) for m in matches],
                          px=np.array([m[2] for m in matches]), T_opt=cur.Get_Pose().copy(), po=(sm["iterations"], sm["termination"]),
                          found=[mp.mnFound for mp in a_mps], bad=[mp.mbBad for mp in a_mps], n_feat=cur.n_features))
        last = cur

    # ---- one call per frame ----
    b_kfs, b_mps = worlds[1]
    trk = tracking.Tracker(cam, ctx=gpu_ctx, max_level=5, min_level=0, max_iters=8, min_tracked=20)
    b_idx = {id(mp): i for i, mp in enumerate(b_mps)}
    last = b_kfs[n_kf - 1]
    for k in range(n_frames):
        cur, n, matches = trk.TrackFrame(imgs[k], last, b_kfs, b_mps)
        r, a = trk.last_result, a_log[k]
        assert n == a["n"] and list(r["stats"]["iters"]) == a["iters"], k
        assert np.array_equal(r["T_run"], a["T_run"]), f"frame {k}: Run pose"
        assert [(m[0], b_idx[id(m[1])], m[3]) for m in matches] == a["matches"], f"frame {k}: match set"
        assert np.array_equal(np.array([m[2] for m in matches]), a["px"]), f"frame {k}: refined pixels"
        assert np.array_equal(cur.Get_Pose(), a["T_opt"]), f"frame {k}: refined pose"
        assert (r["summary"]["iterations"], r["summary"]["termination"]) == a["po"], k
        assert [mp.mnFound for mp in b_mps] == a["found"] and [mp.mbBad for mp in b_mps] == a["bad"], f"frame {k}: map side effects"
        assert cur.n_features == a["n_feat"] and len(matches) >= 60
        last = cur
    assert a_log[-1]["n"] >= 40


@pytest.mark.parametrize("width,height", [(640, 480), (636, 478)])
def test_every_way_the_image_can_arrive_gives_the_same_frame(gpu_ctx, width, height):
    """Level 0 reaches the device on several paths (api.cpp, dsdtm_track_frame step 1): a pageable image is staged through the context's
    pinned block and read from there by ingest_kernel; a pinned, 16-byte aligned image is read by that kernel straight from the
    caller's buffer; a pinned image at an odd address goes through the copy engine; a row-strided image is packed row by row first; an image
    in device memory is read where it is, or copied device to device when it is strided or at an odd address.
    Same bytes on the device, so the same Run pose, match list and refined pose, bit for bit. 636 x 478: the byte count is not a
    multiple of 16 (the kernel's byte tail) and the level widths are not multiples of 8 (one pyrDown launch per level)."""
    import torch
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    cam, kfs, cur, mps = make_world(31, n_points=500, n_kf=2, width=width, height=height)
    ref = kfs[0]
    nf = min(ref.n_features, 200)
    bb = ref.bearing[:nf]
    last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
    last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))
    img = np.ascontiguousarray(cur.mvImg_Pyr[0])
    n = width * height
    assert (n % 16 == 0) == (width == 640)
    pin = torch.empty(n + 64, dtype=torch.uint8).pin_memory()
    pin_s = torch.empty(height * (width + 24), dtype=torch.uint8).pin_memory()
    flat = pin.numpy()
    base = flat.ctypes.data
    assert base % 16 == 0
    aligned = flat[:n].reshape(height, width); aligned[:] = img
    strided = pin_s.numpy().reshape(height, width + 24)[:, :width]; strided[:] = img
    pageable_strided = np.zeros((height, width + 7), np.uint8)[:, :width]; pageable_strided[:] = img

    def run(image):
        r = tracking.track_frame(gpu_ctx, cam, image, 5, last, ref.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps)
        r["frame"].close()
        return r
    want = run(img)
    assert want["n_tracked"] >= 100 and len(want["matches"]) >= 40 and not want["lost"]
    got = {"pinned, aligned": run(aligned), "pinned, row-strided": run(strided), "pageable, row-strided": run(pageable_strided)}
    odd = flat[4:4 + n].reshape(height, width); odd[:] = img           # (shares the buffer with `aligned`, which has been used)
    assert odd.ctypes.data % 16 == 4
    got["pinned, odd address"] = run(odd)
    # an image that already lives in device memory: read from there (contiguous, aligned), or copied device to device (row-strided; odd address)
    dev = torch.from_numpy(img).cuda()
    dev_s = torch.zeros((height, width + 40), dtype=torch.uint8, device="cuda"); dev_s[:, :width] = dev
    dev_o = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"); dev_o[4:4 + n] = dev.reshape(-1)
    torch.cuda.synchronize()
    for name, ptr, stride in (("device, aligned", dev.data_ptr(), width), ("device, row-strided", dev_s.data_ptr(), width + 40),
                              ("device, odd address", dev_o.data_ptr() + 4, width)):
        call = tracking.TrackCall(gpu_ctx, cam, img, 5, last, ref.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps)
        call.desc.image, call.desc.stride = ptr, stride
        got[name] = call.run()
        got[name]["frame"].close()
    assert dev.data_ptr() % 16 == 0
    # an ordinary buffer the caller registered (hipHostRegister: a capture buffer that already exists), page-aligned
    raw = np.zeros(n + 8192, np.uint8)
    off = (-raw.ctypes.data) % 4096
    reg = raw[off:off + n].reshape(height, width); reg[:] = img
    rt = torch.cuda.cudart()
    nreg = (n + 4095) // 4096 * 4096
    if int(rt.cudaHostRegister(reg.ctypes.data, nreg, 2)) == 0:                 # 2 = hipHostRegisterMapped
        try:
            got["registered host buffer"] = run(reg)
        finally:
            rt.cudaHostUnregister(reg.ctypes.data)
    for name, r in got.items():
        assert r["n_tracked"] == want["n_tracked"] and np.array_equal(r["T_run"], want["T_run"]), name
        assert list(r["stats"]["iters"]) == list(want["stats"]["iters"]), name
        assert np.array_equal(r["matches"], want["matches"]) and np.array_equal(r["T_opt"], want["T_opt"]), name
        assert np.array_equal(r["residual_norm"], want["residual_norm"]), name
    # and the frame dsdtm_frame_create_from_image builds (same kernel, from the staging block) is the frame Run aligns against
    al = Sprase_ImgAlign(5, 0, 8, ctx=gpu_ctx, resident_frames=True)
    cur2 = Frame(cam, [img], ref.Get_Pose())
    cur2._device_frame = capi.DeviceFrame.from_image(gpu_ctx, img, 5)
    assert al.Run(cur2, last) == want["n_tracked"] and np.array_equal(cur2.Get_Pose(), want["T_run"])


@pytest.mark.parametrize("nf", [600, 1000])
def test_run_of_a_large_reference_frame_goes_through_the_team_kernel(gpu_ctx, nf):
    """A last frame with 600 / 1000 features: `Run` inside the one-call frame is spread over 3 / 4 compute units (the team kernel, with
    its transparent re-run should a partner wait run out) exactly as dsdtm_sparse_align_frames spreads it: same pose, count and
    iterations, bit for bit; the search and the refinement behind it follow from that pose."""
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    cam, kfs, cur, mps = make_world(17, n_points=1800)
    ref = kfs[0]
    assert ref.n_features >= nf
    bb = ref.bearing[:nf]
    last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
    last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))
    al = Sprase_ImgAlign(5, 0, 8, ctx=gpu_ctx, resident_frames=True)
    c4 = Frame(cam, cur.mvImg_Pyr, last.Get_Pose())
    n4 = al.Run(c4, last)
    r = tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, last.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps[:900])
    assert r["n_tracked"] == n4 and n4 > nf // 2 and list(r["stats"]["iters"]) == list(al.last_stats["iters"])
    assert np.array_equal(r["T_run"], c4.Get_Pose())
    assert len(r["matches"]) > 100 and r["summary"]["n_residual_blocks"] == len(r["matches"])
    r["frame"].close()


@pytest.mark.diag
def test_a_team_timeout_inside_the_one_call_frame_is_re_run_from_the_seed(gpu_ctx_diag):
    """600 reference features: `Run` inside dsdtm_track_frame is a team of three compute units. With the diagnostic switch that keeps the
    last member away, the members' bounded waits run out; the call sees the timeout word after its one wait, puts the seed pose, a zero
    count and cleared statistics back into Run's range ON THE DEVICE (the range lives there since round 6: a second ingest launch), and
    issues the whole chain again with Run on one compute unit. The caller gets the frame an undisturbed call returns, and no error."""
    ctx = gpu_ctx_diag
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    cam, kfs, cur, mps = make_world(17, n_points=1800)
    ref = kfs[0]
    nf = 600
    bb = ref.bearing[:nf]
    last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
    last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))

    def run():
        r = tracking.track_frame(ctx, cam, cur.mvImg_Pyr[0], 5, last, last.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps[:900])
        r["frame"].close()
        return r
    want = run()
    recovered = ctx.lib.dsdtm_debug_recovered_launches
    recovered.restype, recovered.argtypes = C.c_longlong, [C.c_void_p]
    drop = ctx.lib.dsdtm_debug_drop_team_members
    drop.restype, drop.argtypes = None, [C.c_int]
    n0 = recovered(ctx.handle)
    try:
        drop(1)
        got = run()
    finally:
        drop(0)
    assert recovered(ctx.handle) == n0 + 1, "the first attempt did not time out: the switch no longer reaches this launch"
    assert got["n_tracked"] == want["n_tracked"] > nf // 2 and list(got["stats"]["iters"]) == list(want["stats"]["iters"])
    # (one compute unit sums the same partials in another order than three: the pose agrees to rounding, not to the bit)
    H.assert_pose_close(got["T_run"], want["T_run"], H.TIGHT_RAD * 10, H.TIGHT_M * 10, what="Run after the re-run")
    assert [tuple(m)[:2] for m in got["matches"][["cell", "point"]]] == [tuple(m)[:2] for m in want["matches"][["cell", "point"]]]
    assert got["summary"]["iterations"] == want["summary"]["iterations"]
    H.assert_pose_close(got["T_opt"], want["T_opt"], H.TIGHT_RAD * 100, H.TIGHT_M * 100, what="refined pose after the re-run")
    assert np.array_equal(run()["T_run"], want["T_run"])                   # and the next undisturbed frame is the first one again


def test_degenerate_but_valid_shapes(gpu_ctx):
    """What a tracker can legitimately hand over at start-up or in a poor scene: an empty local map, no keyframes at all, a mask that
    blocks the whole image, a cap of ONE match, the largest cell the grid takes. Each call returns OK, the new frame and Run's
    result; the search part is empty where it must be and equals the host replay where it is not."""
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    cam, kfs, cur, mps = make_world(41, n_points=400, n_kf=2)
    ref = kfs[0]
    nf = min(ref.n_features, 150)
    bb = ref.bearing[:nf]
    last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
    last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))
    args = (gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, ref.Get_Pose(), (5, 0, 8, 15), 20)
    want = tracking.track_frame(*args, kfs, mps)
    want["frame"].close()
    assert want["n_tracked"] > 60 and len(want["matches"]) > 30

    def same_run(r):
        return r["n_tracked"] == want["n_tracked"] and np.array_equal(r["T_run"], want["T_run"]) and not r["lost"]
    # an empty local map (with and without keyframes): Run's result, no matches, the refinement has no residual block and keeps Run's pose
    for kf_list in (kfs, []):
        r = tracking.track_frame(*args, kf_list, [])
        assert same_run(r) and r["n_in_grid"] == 0 and len(r["matches"]) == 0
        # (the refinement re-assembles the pose from its parameter block, log then exp, as src/Optimizer.cpp:35-37,78 does: equal to rounding)
        assert r["summary"]["n_residual_blocks"] == 0 and np.allclose(r["T_opt"], r["T_run"], rtol=0, atol=1e-14)
        r["frame"].close()
    # a mask that blocks everything: every candidate is skipped (:96)
    r = tracking.track_frame(*args, kfs, mps, mask=np.zeros((cam.height, cam.width), np.uint8))
    assert same_run(r) and r["n_in_grid"] == want["n_in_grid"] and len(r["matches"]) == 0 and np.allclose(r["T_opt"], r["T_run"], rtol=0, atol=1e-14)
    r["frame"].close()
    # a cap of one match: the walk's FIRST success, and the refinement on that single feature
    r = tracking.track_frame(*args, kfs, mps, max_matches=1)
    assert same_run(r) and len(r["matches"]) == 1 and r["matches"][0] == want["matches"][0] and r["summary"]["n_residual_blocks"] == 1
    r["frame"].close()
    # the largest cell (127 px: a 6 x 4 grid, every candidate's neighbourhood is most of the image -> the full scan)
    mps_h = copy.deepcopy(mps)
    Config.Set("Camera.CellSize", 127)
    try:
        c4 = Frame(cam, cur.mvImg_Pyr, want["T_run"])
        s = search.LocalPointSearch(cam, ctx=gpu_ctx, resident_frames=True)
        s.ResetGrid()
        for mp in mps_h:
            if not mp.IsBad():
                s.ReprojectPoint(c4, mp)
        idx = {id(mp): i for i, mp in enumerate(mps_h)}
        want127 = [(g[0], idx[id(g[1])], float(g[2][0]), float(g[2][1]), g[3]) for g in s.SearchLocalPoints(c4, kfs)]
    finally:
        Config.Set("Camera.CellSize", 25)
    r = tracking.track_frame(*args, kfs, mps, cell_size=127)
    m = r["matches"]
    got127 = [(int(m["cell"][k]), int(m["point"][k]), float(m["px"][k][0]), float(m["px"][k][1]), int(m["level"][k])) for k in range(len(m))]
    assert same_run(r) and got127 == want127 and 1 <= len(got127) <= 24
    r["frame"].close()


def test_lost_frame_skips_search_and_refinement(gpu_ctx):
    """Run's count below Tracking's threshold (src/Tracking.cpp:208: < 20 => Lost): nothing after Run is computed — no matches,
    T_opt = T_run — and the new frame is still handed over."""
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    cam, kfs, cur, mps = make_world(11, n_points=300)
    ref = kfs[0]
    nf = 40
    bb = ref.bearing[:nf]
    last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
    last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))
    r = tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, last.Get_Pose(), (5, 0, 8, 15), 1000, kfs, mps)
    assert r["lost"] and 0 < r["n_tracked"] <= nf and len(r["matches"]) == 0 and np.array_equal(r["T_opt"], r["T_run"])
    r2 = tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, last.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps)
    assert not r2["lost"] and r2["n_tracked"] == r["n_tracked"] and np.array_equal(r2["T_run"], r["T_run"]) and len(r2["matches"]) > 50
    r["frame"].close(); r2["frame"].close()


def test_argument_checks(gpu_ctx):
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5)
    cam, kfs, cur, mps = make_world(11, n_points=50)
    last = Frame(cam, cur.mvImg_Pyr, cur.Get_Pose())
    flat = tracking.flatten_local_map(kfs, mps)
    ok = lambda **kw: tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, last, cur.Get_Pose(), (5, 0, 8, 15), 0, kfs, mps, **kw)
    ok()["frame"].close()
    for kw in (dict(cell_size=0), dict(cell_size=3), dict(max_matches=0), dict(max_matches=257), dict(max_pyr_levels=2), dict(max_pyr_levels=9)):
        with pytest.raises(capi.DsdtmError) as e:
            ok(**kw)
        assert e.value.status == capi.ERR_INVALID, kw
    bad = dict(flat)
    bad["okf"] = flat["okf"].copy(); bad["okf"][0] = 99            # an observation that names a keyframe outside the list
    with pytest.raises(capi.DsdtmError):
        ok(flat=bad)
    bad = dict(flat)
    bad["off"] = flat["off"].copy(); bad["off"][3] = bad["off"][2] - 1
    with pytest.raises(capi.DsdtmError):
        ok(flat=bad)
    # a reference frame of another geometry
    small = Frame(synth.Camera.tum(320, 240), synth.build_pyramid(cur.mvImg_Pyr[1], 5), cur.Get_Pose())
    with pytest.raises(capi.DsdtmError):
        tracking.track_frame(gpu_ctx, cam, cur.mvImg_Pyr[0], 5, small, cur.Get_Pose(), (5, 0, 8, 15), 0, kfs, mps)
    # and the context still works
    ok()["frame"].close()


def test_two_trackers_on_one_gpu_from_two_threads(gpu_ctx):
    """Live tracking of one sequence is a dependent chain: more throughput = more trackers (replicas), each with its own context
    (DESIGN §7). Two contexts on device 0 driven from two host threads at once, 12 frames each on worlds of their own: every call
    returns what the same call returns alone (the contexts share nothing: staging, streams, frame pools, timeout words)."""
    import threading
    Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
    worlds = []
    for seed in (31, 32):
        cam, kfs, cur, mps = make_world(seed, n_points=700)
        ref = kfs[0]
        nf = 250
        bb = ref.bearing[:nf]
        last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
        last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))
        worlds.append((cam, kfs, cur, mps, last))

    def run(ctx, w, n):
        cam, kfs, cur, mps, last = w
        call = tracking.TrackCall(ctx, cam, cur.mvImg_Pyr[0], 5, last, last.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps)
        out = []
        for _ in range(n):
            r = call.run()
            out.append((r["T_run"].copy(), r["n_tracked"], r["matches"].copy(), r["T_opt"].copy(), r["summary"]["iterations"]))
            r["frame"].close()
        return out

    alone = [run(gpu_ctx, w, 1)[0] for w in worlds]
    ctxs = [capi.Context(0), capi.Context(0)]
    res = [None, None]
    th = [threading.Thread(target=lambda k=k: res.__setitem__(k, run(ctxs[k], worlds[k], 12))) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert res[k] is not None and len(res[k]) == 12
        for T_run, n, m, T_opt, its in res[k]:
            assert np.array_equal(T_run, alone[k][0]) and n == alone[k][1] and np.array_equal(m, alone[k][2])
            assert np.array_equal(T_opt, alone[k][3]) and its == alone[k][4]
        assert len(alone[k][2]) > 60
    for c in ctxs:
        c.close()
