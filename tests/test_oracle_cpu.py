"""CPU suite (no GPU): pins the oracle — against the committed golden fixtures, against an
independent numpy restatement of the reference text, against analytic ground truth and
textbook identities for the restated third-party arithmetic (Sophus SE3, Eigen LDLT, cv::pyrDown).

The reference holds no golden vectors for this path: parity is "unpinned" by
the reference, so these self-checks are what stands behind the oracle."""
import ctypes as C

import numpy as np
import pytest

from dsdtm_amd import synth
from tests import helpers as H
from tests import np_restatement as NP
from tests.conftest import cached_scene

dp = C.POINTER(C.c_double)


def _arr(a):
    return np.ascontiguousarray(a, np.float64)


# ---------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("name", ["sparse_align_a.npz", "sparse_align_b.npz"])
def test_oracle_reproduces_golden_sparse_align(oracle, name):
    g = H.GoldenScene(H.golden_path(name))
    T, n, st = oracle.sparse_align(g, *g.params)
    assert np.array_equal(T, g.d["out_T"])
    assert n == int(g.d["out_n"])
    assert st["iters"] == list(g.d["out_iters"]) and st["exit_code"] == list(g.d["out_exit"])
    assert st["n_ref"] == list(g.d["out_nref"]) and st["n_vis"] == list(g.d["out_nvis"])
    assert np.array_equal(np.array(st["chi2"]), g.d["out_chi2"])
    # and the golden result is meaningful: close to the synthetic ground truth
    ea, et = synth.pose_error(T, g.T_cur_w_true)
    assert ea < 5e-4 and et < 1e-3


def test_oracle_reproduces_golden_align2d_pyr_warp(oracle):
    d = np.load(H.golden_path("align2d_pyr_warp.npz"))
    pyr = [d["tex"], oracle.pyrdown(d["tex"])]
    pyr.append(oracle.pyrdown(pyr[1]))
    assert np.array_equal(pyr[1], d["pyr1"]) and np.array_equal(pyr[2], d["pyr2"])
    conv, px = oracle.align2d_batch(pyr, d["patch_border"], d["patch"], d["level"], d["px0"], 10)
    assert np.array_equal(conv, d["out_conv"]) and np.array_equal(px, d["out_px"])
    c = d["cam"]
    cam = synth.Camera(c[0], c[1], c[2], c[3], c[4], int(c[5]), int(c[6]))
    aff, sl, wb, wp = oracle.warp_patches([pyr, pyr], cam, d["T_kf"], d["T_cur"], d["cand_kf"], d["ref_px"],
                                          d["ref_level"], d["ref_bearing"], d["p_world"], 2)
    assert np.array_equal(aff, d["out_affine"]) and np.array_equal(sl, d["out_search_level"])
    assert np.array_equal(wb, d["out_warp_border"]) and np.array_equal(wp, d["out_warp_patch"])


# ---------------------------------------------------------------- independent restatement
@pytest.mark.parametrize("kw,params", [
    (dict(width=320, height=240, levels=3, n_patches=120, seed=41, margin=12), (3, 0, 10)),
    (dict(width=320, height=240, levels=4, n_patches=200, seed=42, margin=12, frac_uninitial=0.1), (4, 1, 30)),
    (dict(width=640, height=480, levels=4, n_patches=300, seed=0xD5D7), (4, 0, 10)),      # BASELINE config 2
])
def test_oracle_matches_independent_numpy_restatement(oracle, kw, params):
    sc = cached_scene(**kw)
    To, no, so = oracle.sparse_align(sc, *params)
    Tn, nn, itn = NP.sparse_align(sc, *params)
    ang, dt = synth.pose_error(To, Tn)
    assert ang < 1e-9 and dt < 1e-9, (ang, dt)
    assert no == nn and so["iters"] == itn


def test_oracle_no_visible_patch_matches_restatement(oracle):
    sc = cached_scene(width=320, height=240, levels=3, n_patches=60, seed=6, margin=12)
    T_seed = synth.se3_exp([0, 0, 0, 0, 1.2, 0])[:3] @ np.vstack([sc.T_ref_w, [0, 0, 0, 1]])
    To, no, so = oracle.sparse_align(sc, 3, 0, 10, T_seed=T_seed)
    Tn, nn, itn = NP.sparse_align(sc, 3, 0, 10, T_seed=T_seed)
    assert no == nn == 0 and so["iters"] == itn
    assert synth.pose_error(To, T_seed)[0] < 1e-12 and synth.pose_error(To, Tn)[0] < 1e-12


def test_config2_recovers_ground_truth(oracle):
    """Known synthetic motion recovered to < 2e-4 rad / 3e-4 m (SURVEY.md §8c analytic check)."""
    sc = cached_scene(width=640, height=480, levels=4, n_patches=300, seed=0xD5D7)
    T, n, st = oracle.sparse_align(sc, 4, 0, 10)
    ea, et = synth.pose_error(T, sc.T_cur_w_true)
    assert n == 300 and ea < 2e-4 and et < 3e-4
    assert all(1 <= i <= 10 for i in st["iters"][:4])


def test_min_fts_rule(oracle):
    sc = cached_scene(width=320, height=240, levels=3, n_patches=10, seed=5, margin=12)
    T, n, st = oracle.sparse_align(sc, 3, 0, 10, min_fts=15)
    assert n == 0 and np.array_equal(T, sc.T_cur_w_seed) and st["iters"] == [0] * 8


def test_align2d_matches_independent_restatement(oracle):
    tex = np.clip(np.rint(synth.make_texture(120, 160, 3)), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(2)
    for _ in range(12):
        c = (rng.uniform(12, 148), rng.uniform(12, 108))
        pb, p = H.make_border_patches(tex, [c])
        px0 = np.array([c[0] + rng.uniform(-1.5, 1.5), c[1] + rng.uniform(-1.5, 1.5)])
        oko, pxo = oracle.align2d(tex, pb[0], p[0], 10, px0)
        okn, pxn = NP.align2d(tex, pb[0], p[0], 10, px0)
        assert oko == okn
        assert np.allclose(pxo, pxn, atol=2e-3)       # float32; Hinv by cofactors vs LU


def test_align2d_reference_scenario(oracle):
    """Test/test_Feature_alignment.cpp:56-81: start 1.1/0.8 px off, converge to ~1e-2 px."""
    tex = np.clip(np.rint(synth.make_texture(240, 320, 3)), 0, 255).astype(np.uint8)
    pb, p = H.make_border_patches(tex, [(130.2, 120.3)])
    ok, px = oracle.align2d(tex, pb[0], p[0], 10, np.array([130.2 - 1.1, 120.3 - 0.8]))
    assert ok and np.hypot(*(px - [130.2, 120.3])) < 0.05


# ---------------------------------------------------------------- restated third-party arithmetic
def test_se3_identities(oracle):
    lib = oracle.load()
    rng = np.random.default_rng(0)
    for i in range(50):
        xi = rng.standard_normal(6) * (1e-12 if i % 7 == 0 else 0.5)
        E, Ei, I = oracle.OracleSE3(), oracle.OracleSE3(), oracle.OracleSE3()
        lib.oracle_se3_exp(_arr(xi).ctypes.data_as(dp), C.byref(E))
        T = np.zeros(12)
        lib.oracle_se3_to_rt(C.byref(E), T.ctypes.data_as(dp))
        assert np.allclose(T.reshape(3, 4), synth.se3_exp(xi)[:3], atol=1e-13)       # vs closed form
        lib.oracle_se3_inverse(C.byref(E), C.byref(Ei))
        lib.oracle_se3_mul(C.byref(E), C.byref(Ei), C.byref(I))
        assert np.allclose(list(I.q), [1, 0, 0, 0], atol=1e-14) and np.allclose(list(I.t), 0, atol=1e-14)
        E2 = oracle.OracleSE3()
        lib.oracle_se3_from_rt(T.ctypes.data_as(dp), C.byref(E2))                     # round trip through [R|t]
        s = np.sign(E2.q[0] * E.q[0]) or 1.0
        assert np.allclose(np.array(list(E2.q)) * s, list(E.q), atol=1e-13)
        p = rng.standard_normal(3)
        out = np.zeros(3)
        lib.oracle_se3_act(C.byref(E), _arr(p).ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert np.allclose(out, T.reshape(3, 4)[:, :3] @ p + T.reshape(3, 4)[:, 3], atol=1e-13)


def test_ldlt_solves_and_handles_rank_deficiency(oracle):
    lib = oracle.load()
    rng = np.random.default_rng(1)
    for i in range(40):
        J = rng.standard_normal((30, 6)) * np.array([20, 20, 20, 90, 90, 90])
        Hm = J.T @ J
        b = rng.standard_normal(6)
        x = np.zeros(6)
        lib.oracle_ldlt6_solve(_arr(Hm.reshape(36)).ctypes.data_as(dp), _arr(b).ctypes.data_as(dp), x.ctypes.data_as(dp))
        assert np.allclose(Hm @ x, b, rtol=1e-8, atol=1e-8 * np.abs(b).max())
    # zero matrix: Eigen's pseudo-inverse solve returns 0 (not NaN) -> quirk Q11 path
    x = np.ones(6)
    lib.oracle_ldlt6_solve(_arr(np.zeros(36)).ctypes.data_as(dp), _arr(np.zeros(6)).ctypes.data_as(dp), x.ctypes.data_as(dp))
    assert np.array_equal(x, np.zeros(6))
    # NaN in H propagates to x (the reference's isnan(x(0)) guard, :321)
    Hn = np.eye(6); Hn[0, 0] = np.nan
    lib.oracle_ldlt6_solve(_arr(Hn.reshape(36)).ctypes.data_as(dp), _arr(np.ones(6)).ctypes.data_as(dp), x.ctypes.data_as(dp))
    assert np.isnan(x[0])


def test_jacobian_ba_is_minus_projection_derivative(oracle):
    """SVO sign convention: J = -d(x/z, y/z)/d(xi) for T <- exp(xi) * T."""
    lib = oracle.load()
    rng = np.random.default_rng(3)
    for _ in range(10):
        X = rng.uniform([-1, -1, 1], [1, 1, 4])
        J = np.zeros(12)
        lib.oracle_jacobian_ba(_arr(X).ctypes.data_as(dp), J.ctypes.data_as(dp))
        num = np.zeros((2, 6))
        for k in range(6):
            e = np.zeros(6); e[k] = 1e-6
            Pp = synth.se3_exp(e)[:3, :3] @ X + synth.se3_exp(e)[:3, 3]
            Pm = synth.se3_exp(-e)[:3, :3] @ X + synth.se3_exp(-e)[:3, 3]
            num[:, k] = (Pp[:2] / Pp[2] - Pm[:2] / Pm[2]) / 2e-6
        assert np.allclose(J.reshape(2, 6), -num, atol=1e-6)


@pytest.mark.parametrize("shape", [(480, 640), (241, 323), (60, 80), (5, 7), (2, 3), (1, 9), (7, 1)])
def test_pyrdown_matches_numpy_and_closed_forms(oracle, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    out = oracle.pyrdown(img)
    assert out.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
    assert np.array_equal(out, synth.pyrdown_u8(img))
    const = np.full(shape, 137, np.uint8)
    assert np.array_equal(oracle.pyrdown(const), np.full(out.shape, 137, np.uint8))     # kernel sums to 256
    if min(shape) >= 9:
        imp = np.zeros(shape, np.uint8); imp[4, 4] = 255                               # impulse: 6*6*255/256 -> 36
        assert oracle.pyrdown(imp)[2, 2] == (36 * 255 + 128) >> 8
        assert oracle.pyrdown(imp)[1, 1] == (1 * 255 + 128) >> 8


def test_warp_integer_division_quirk(oracle):
    """W1: search level >= 1 -> every sample collapses onto the reference pixel (constant patch)."""
    d = np.load(H.golden_path("align2d_pyr_warp.npz"))
    sl, wb = d["out_search_level"], d["out_warp_border"]
    assert ((wb.max(axis=1) == wb.min(axis=1)) == (sl >= 1)).all() or (sl == 0).all()


def test_interpolated_reference_intensities_are_exact_in_fp64():
    """What the kernels rely on when they rebuild the reference grid (workspace kernel, parity across kernel shapes):
    the bilinear value w00 a + w01 b + w10 c + w11 d of four bytes at a reference-side subpixel position is EXACT in
    FP64 — the position is a float32 pixel >= 3 * 2^level (border test, src/Sprase_ImageAlign.cpp:95-100) times 2^-level,
    so each subpixel offset has at most 22 fractional bits — hence the same bits in any order of operations."""
    from fractions import Fraction
    rng = np.random.default_rng(5)
    for level in range(5):
        scale = np.float64(np.float32(1.0) / np.float32(1 << level))
        for _ in range(400):
            px = np.float32(rng.uniform(3 * (1 << level), 640.0))
            py = np.float32(rng.uniform(3 * (1 << level), 480.0))
            x, y = np.float64(px) * scale, np.float64(py) * scale
            su, sv = x - np.floor(x), y - np.floor(y)
            w = [(1.0 - su) * (1.0 - sv), su * (1.0 - sv), (1.0 - su) * sv, su * sv]
            exact_w = [Fraction(1 - Fraction(su)) * Fraction(1 - Fraction(sv)), Fraction(su) * (1 - Fraction(sv)),
                       (1 - Fraction(su)) * Fraction(sv), Fraction(su) * Fraction(sv)]
            assert all(Fraction(float(a)) == b for a, b in zip(w, exact_w))          # the weights themselves are exact
            q = rng.integers(0, 256, 4)
            exact = sum(Fraction(int(q[i])) * exact_w[i] for i in range(4))
            fwd = ((w[0] * q[0] + w[1] * q[1]) + w[2] * q[2]) + w[3] * q[3]
            rev = ((w[3] * q[3] + w[2] * q[2]) + w[1] * q[1]) + w[0] * q[0]
            assert Fraction(float(fwd)) == exact and Fraction(float(rev)) == exact


def test_restated_third_party_arithmetic_against_scipy(oracle):
    """The third-party pieces of the path are not under /root/reference (Sophus, Eigen, OpenCV) and are
    restated in the oracle from their published definitions. Independent implementations to hold them against: scipy's
    matrix exponential of the 4x4 twist (SE3::exp), scipy's Rotation (quaternion -> matrix, composition), LAPACK through
    numpy (H x = b), and scipy.ndimage's separable correlation with mirrored borders (cv::pyrDown = [1 4 6 4 1]^2 / 256,
    BORDER_REFLECT_101, rounded half up, every second pixel)."""
    from scipy.linalg import expm
    from scipy.ndimage import correlate1d
    from scipy.spatial.transform import Rotation
    lib = oracle.load()
    rng = np.random.default_rng(42)
    for i in range(40):
        xi = rng.standard_normal(6) * (1e-9 if i % 8 == 0 else (0.05 if i % 2 else 1.5))
        u, w = xi[:3], xi[3:]
        tw = np.zeros((4, 4))
        tw[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
        tw[:3, 3] = u
        E = oracle.OracleSE3()
        lib.oracle_se3_exp(_arr(xi).ctypes.data_as(dp), C.byref(E))
        T = np.zeros(12)
        lib.oracle_se3_to_rt(C.byref(E), T.ctypes.data_as(dp))
        assert np.allclose(T.reshape(3, 4), expm(tw)[:3], atol=1e-12)
        q = np.array(list(E.q))                                                        # (w, x, y, z)
        assert np.allclose(Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(), T.reshape(3, 4)[:, :3], atol=1e-13)
        xj = rng.standard_normal(6) * 0.3
        Ej, Ek = oracle.OracleSE3(), oracle.OracleSE3()
        lib.oracle_se3_exp(_arr(xj).ctypes.data_as(dp), C.byref(Ej))
        lib.oracle_se3_mul(C.byref(E), C.byref(Ej), C.byref(Ek))
        Tk = np.zeros(12)
        lib.oracle_se3_to_rt(C.byref(Ek), Tk.ctypes.data_as(dp))
        twj = np.zeros((4, 4))
        twj[:3, :3] = [[0, -xj[5], xj[4]], [xj[5], 0, -xj[3]], [-xj[4], xj[3], 0]]
        twj[:3, 3] = xj[:3]
        assert np.allclose(Tk.reshape(3, 4), (expm(tw) @ expm(twj))[:3], atol=1e-12)
        J = rng.standard_normal((40, 6)) * np.array([20, 20, 20, 90, 90, 90])
        Hm, b, x = J.T @ J, rng.standard_normal(6), np.zeros(6)
        lib.oracle_ldlt6_solve(_arr(Hm.reshape(36)).ctypes.data_as(dp), _arr(b).ctypes.data_as(dp), x.ctypes.data_as(dp))
        assert np.allclose(x, np.linalg.solve(Hm, b), rtol=1e-7, atol=1e-12)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    for shape in [(48, 64), (37, 53), (9, 9), (5, 12)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        acc = correlate1d(correlate1d(img.astype(np.int64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
        want = ((acc + 128) >> 8)[::2, ::2].astype(np.uint8)
        assert np.array_equal(oracle.pyrdown(img), want)


def _test1_scene(n_features=300, levels=4):
    """The reference's own test image (Thirdparty/fast/test/data/test1.png, committed as data in the FAST fixtures) with the
    corners ITS FAST build found at barrier 20 (`test1_b20`: pinned reference output) as features, seen again through a
    plane-warp at the Config/Rpg_uzh.yaml intrinsics."""
    d = np.load(H.golden_path("fast_reference.npz"))
    img = np.ascontiguousarray(d["test1"])
    corners = d["test1_b20"][:, :2].astype(np.float32)                   # (x, y) of the reference detector's corners
    rng = np.random.default_rng(17)
    inside = (corners[:, 0] > 30) & (corners[:, 0] < 752 - 30) & (corners[:, 1] > 30) & (corners[:, 1] < 480 - 30)
    px = corners[inside][rng.choice(int(inside.sum()), n_features, replace=False)]
    cam = synth.Camera(315.5, 315.5, 376.0, 240.0, 315.5, 752, 480)
    T_cr = synth.se3_exp((0.012, -0.007, 0.005, 0.004, -0.003, 0.006))
    depth = 2.0
    cur = synth.warp_plane(img.astype(np.float64), cam, T_cr, depth)
    bearing = synth.bearing_from_px(cam, px)
    T_ref = synth.random_pose(rng)
    X_r = bearing * (depth / bearing[:, 2:3])
    p_world = (X_r - T_ref[:, 3]) @ T_ref[:, :3]
    T4 = np.vstack([T_ref, [0, 0, 0, 1]])
    return synth.AlignScene(cam, synth.build_pyramid(img, levels), synth.build_pyramid(cur, levels), px, bearing, p_world,
                            np.ones(len(px), np.uint8), T_ref.copy(), T_ref.copy(), (T_cr @ T4)[:3], depth)


def test_oracle_matches_numpy_restatement_on_the_reference_image(oracle):
    """The two independent CPU restatements agree on REAL image statistics too: Run on test1 at the reference detector's own
    corners (clustered on structure, some on saturated edges) — same pose to 1e-9, same iteration counts; and the warp is
    recovered."""
    sc = _test1_scene()
    To, no, so = oracle.sparse_align(sc, 4, 0, 10)
    Tn, nn, itn = NP.sparse_align(sc, 4, 0, 10)
    ang, dt = synth.pose_error(To, Tn)
    assert ang < 1e-9 and dt < 1e-9, (ang, dt)
    assert no == nn and so["iters"] == itn and no > 250
    ea, et = synth.pose_error(To, sc.T_cur_w_true)
    assert ea < 2e-3 and et < 5e-3, (ea, et)


def test_align2d_restatements_agree_on_the_reference_image(oracle):
    """Align2DGaussNewton at the reference detector's corners of test1: oracle and numpy restatement, pixel for pixel."""
    d = np.load(H.golden_path("fast_reference.npz"))
    img = np.ascontiguousarray(d["test1"])
    corners = d["test1_b75"][:, :2].astype(np.float64)                   # the 167 corners of the reference's own test
    rng = np.random.default_rng(4)
    n_ok = 0
    for c in corners[(corners[:, 0] > 14) & (corners[:, 0] < 752 - 14) & (corners[:, 1] > 14) & (corners[:, 1] < 480 - 14)][:40]:
        pb, p = H.make_border_patches(img, [tuple(c + rng.uniform(0, 1, 2))])
        px0 = c + rng.uniform(-1.5, 1.5, 2)
        oko, pxo = oracle.align2d(img, pb[0], p[0], 10, px0)
        okn, pxn = NP.align2d(img, pb[0], p[0], 10, px0)
        assert oko == okn and np.allclose(pxo, pxn, rtol=0, atol=1e-4, equal_nan=True), (c, pxo, pxn)
        n_ok += oko
    assert n_ok >= 25
