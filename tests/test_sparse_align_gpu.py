"""GPU parity of Sprase_ImgAlign::Run (HIP path through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from dsdtm_amd import synth
from tests import helpers as H
from tests.conftest import cached_scene

pytestmark = pytest.mark.gpu


def test_device_building_blocks(gpu_ctx, oracle):
    """DPP wave reduction, pivoted LDLT, SE(3) exp/mul on the device vs the oracle's."""
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
        xi = rng.standard_normal(6) * (1e-12 if i % 5 == 0 else 0.3)
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


@pytest.mark.parametrize("n_patches,size", [(330, (320, 240)), (448, (320, 240)), (1000, (640, 480)), (2000, (640, 480))])
def test_large_patch_counts(gpu_ctx, oracle, n_patches, size):
    """448-lane register kernel and the workspace kernel (configs 3 and 5 patch counts)."""
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


def test_identity_motion_converges_immediately(gpu_ctx, oracle):
    sc = cached_scene(width=320, height=240, levels=3, n_patches=150, seed=11, xi=(0, 0, 0, 0, 0, 0), margin=12)
    Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=gpu_ctx)
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    H.assert_pose_close(Tg, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10)
    ea, et = synth.pose_error(Tg, sc.T_cur_w_true)
    assert ea < 1e-4 and et < 2e-4 and sg["iters"] == so["iters"]
