"""CPU side of Optimizer::PoseOptimization: the C restatement
(oracle/pose_opt_oracle.c) against an independent numpy restatement, an independent minimiser, the
committed vectors, and the quirks of the reference it has to keep."""
import numpy as np
import pytest
from scipy.optimize import minimize

from dsdtm_amd import capi, synth
from tests import helpers as H
from tests import pose_opt_restatement as R

CASES = [dict(seed=1, n=200, max_level=3), dict(seed=2, n=200, max_level=0), dict(seed=3, n=500, max_level=4, outlier_frac=0.2),
         dict(seed=4, n=30, max_level=2), dict(seed=5, n=150, max_level=3, seed_t=0.12, seed_w=0.1), dict(seed=6, n=7, max_level=1, unused_frac=0.0)]


@pytest.mark.parametrize("kw", CASES)
def test_c_restatement_equals_numpy_restatement(oracle, kw):
    """Same iterations, same decisions, same trust-region radii — two implementations that share no code
    (Sophus-style quaternions + hand-written Householder QR vs scipy Rotation + LAPACK QR)."""
    P = synth.make_pose_problem(**kw)
    T, rn, sm, tr = oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=0, trace=True)
    Tn, rnn, smn, trn = R.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed)
    assert (sm["iterations"], sm["successful_steps"], sm["termination"]) == (smn["iterations"], smn["successful_steps"], smn["termination"])
    ang, dt = synth.pose_error(T, Tn)
    assert ang < 1e-12 and dt < 1e-12
    assert np.allclose(rn, rnn, rtol=0, atol=1e-13)
    k = min(len(tr), len(trn))
    assert np.allclose(tr[:k, :2], trn[:k, :2], rtol=1e-9)          # cost and radius per iteration
    assert abs(sm["final_cost"] - smn["final_cost"]) <= 1e-12 * sm["final_cost"]


@pytest.mark.parametrize("kw", CASES)
def test_normal_equation_form_equals_qr_form(oracle, kw):
    """The form the HIP kernel computes (6x6 normal equations, Cholesky) against Ceres' DENSE_QR form."""
    P = synth.make_pose_problem(**kw)
    T0, rn0, sm0 = oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=0)
    T1, rn1, sm1 = oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=1)
    assert (sm0["iterations"], sm0["successful_steps"], sm0["termination"]) == (sm1["iterations"], sm1["successful_steps"], sm1["termination"])
    ang, dt = synth.pose_error(T0, T1)
    assert ang < 1e-10 and dt < 1e-10
    assert np.allclose(rn0, rn1, rtol=0, atol=1e-11)


def test_result_is_a_minimum_of_the_robust_cost(oracle):
    """With every feature on level 0 the reference's Jacobian is the true one, so the minimiser must end
    at a stationary point of 1/2 sum log(1 + |r|^2): an independent BFGS started there cannot improve the
    cost by more than the function tolerance the solver stops at."""
    P = synth.make_pose_problem(21, n=250, max_level=0, outlier_frac=0.1)
    T, rn, sm = oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed)
    assert sm["termination"] in (capi.PO_FUNCTION_TOLERANCE, capi.PO_PARAMETER_TOLERANCE, capi.PO_GRADIENT_TOLERANCE)
    T4 = np.vstack([T, [0, 0, 0, 1]])
    cost = lambda xi: R.robust_cost((synth.se3_exp(xi) @ T4)[:3], P.bearing, P.p_world, P.level, P.use)
    assert abs(cost(np.zeros(6)) - sm["final_cost"]) <= 1e-12 * sm["final_cost"]
    best = minimize(cost, np.zeros(6), method="BFGS", options=dict(gtol=1e-12))
    assert best.fun <= sm["final_cost"] and sm["final_cost"] - best.fun <= 2e-6 * sm["final_cost"]
    assert sm["final_cost"] < 0.9 * sm["initial_cost"]
    # and it moved towards the truth
    assert sum(synth.pose_error(T, P.T_true)) < 0.3 * sum(synth.pose_error(P.T_seed, P.T_true))


def test_committed_vectors(oracle):
    g = np.load(H.golden_path("pose_opt.npz"))
    for k in range(int(g["n_cases"])):
        for ls, tol in ((0, 1e-13), (1, 1e-10)):
            T, rn, sm = oracle.pose_optimization(g[f"bearing{k}"], g[f"p_world{k}"], g[f"level{k}"], g[f"use{k}"], g[f"T_seed{k}"],
                                                 linear_solver=ls)
            assert [sm["iterations"], sm["successful_steps"], sm["termination"], sm["n_residual_blocks"]] == list(g[f"summary{k}"])
            ang, dt = synth.pose_error(T, g[f"T_out{k}"])
            assert ang <= tol and dt <= tol
            assert np.allclose(rn, g[f"residual_norm{k}"], rtol=0, atol=10 * tol)
            assert np.allclose([sm["initial_cost"], sm["final_cost"]], g[f"cost{k}"], rtol=1e-10)


def test_quirks_of_the_reference(oracle):
    cam = synth.Camera.tum()
    P = synth.make_pose_problem(31, n=60, max_level=0, noise_px=0.0, outlier_frac=0.0, unused_frac=0.0)
    # (1) the residual is divided by 1 << level (include/Optimizer.h:162), the Jacobian is not (:176-189):
    #     same geometry on level 2 -> a quarter of the residual norm at the seed pose
    z = np.zeros(60, np.int32)
    _, rn0, _ = oracle.pose_optimization(P.bearing, P.p_world, z, P.use, P.T_seed, max_iterations=0)
    _, rn2, _ = oracle.pose_optimization(P.bearing, P.p_world, z + 2, P.use, P.T_seed, max_iterations=0)
    assert np.allclose(rn2, rn0 / 4, rtol=1e-15)
    # ... and with exact data the solver still reaches the true pose on level 0, quadratically
    T, rn, sm = oracle.pose_optimization(P.bearing, P.p_world, z, P.use, P.T_seed)
    assert max(synth.pose_error(T, P.T_true)) < 1e-6 and sm["iterations"] <= 8   # f32 bearings: ~1e-7
    # (2) features without a usable map point add no block; the norms come out in block order
    use = P.use.copy(); use[::3] = 0
    _, rnu, smu = oracle.pose_optimization(P.bearing, P.p_world, z, use, P.T_seed, max_iterations=0)
    assert smu["n_residual_blocks"] == int(use.sum()) and np.array_equal(rnu, rn0[use.astype(bool)])
    # (3) no block at all: nothing to solve; the pose goes through log/exp once (:35-37, :78)
    _, rne, sme = oracle.pose_optimization(P.bearing, P.p_world, z, use * 0, P.T_seed)
    Te, _, _ = oracle.pose_optimization(P.bearing, P.p_world, z, use * 0, P.T_seed)
    assert sme["termination"] == capi.PO_NO_RESIDUALS and len(rne) == 0 and max(synth.pose_error(Te, P.T_seed)) < 1e-15
    Tn, _, smn = oracle.pose_optimization(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.uint8), P.T_seed)
    assert smn["termination"] == capi.PO_NO_RESIDUALS and max(synth.pose_error(Tn, P.T_seed)) < 1e-15
    # (4) a point on the camera plane (z = 0): Ceres refuses the initial evaluation, parameters stay
    pw = P.p_world.copy()
    R_, t_ = P.T_seed[:, :3], P.T_seed[:, 3]
    pw[5] = R_.T @ (np.array([0.3, -0.2, 0.0]) - t_)
    pc = R_ @ pw[5] + t_
    pw[5] -= R_.T @ np.array([0, 0, pc[2]])          # exactly zero depth after rounding is not guaranteed ...
    Tz, _, smz = oracle.pose_optimization(P.bearing, pw, z, P.use, P.T_seed)
    if smz["termination"] == capi.PO_EVALUATION_FAILED:   # ... so only assert the consequence when it is
        assert max(synth.pose_error(Tz, P.T_seed)) < 1e-15
    # (5) max_num_iterations is honoured
    P2 = synth.make_pose_problem(32, n=200, max_level=3)
    _, _, sm3 = oracle.pose_optimization(P2.bearing, P2.p_world, P2.level, P2.use, P2.T_seed, max_iterations=3)
    assert sm3["iterations"] == 3 and sm3["termination"] == capi.PO_MAX_ITERATIONS


def test_building_blocks(oracle):
    import ctypes as C
    from scipy.spatial.transform import Rotation
    lib = oracle.load()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(5)
    for _ in range(50):
        w = rng.standard_normal(3)
        w *= rng.choice([1e-12, 1e-3, 0.5, 3.0]) / np.linalg.norm(w)      # |w| < pi: the principal branch
        q = Rotation.from_rotvec(w).as_quat()                  # x, y, z, w
        qs = np.array([q[3], q[0], q[1], q[2]])
        out = np.zeros(3)
        lib.oracle_so3_log(qs.ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert np.allclose(out, w, rtol=1e-12, atol=1e-15)
        x = np.concatenate([rng.standard_normal(3), w]); d = rng.standard_normal(6) * 0.1
        xp = np.zeros(6)
        lib.oracle_pose_plus(x.ctypes.data_as(dp), d.ctypes.data_as(dp), xp.ctypes.data_as(dp))
        assert np.allclose(xp, R.plus(x, d), rtol=1e-11, atol=1e-13)
        A = rng.standard_normal((9, 6)); M = A.T @ A + 1e-3 * np.eye(6); v = rng.standard_normal(6)
        y = np.zeros(6)
        assert lib.oracle_chol6_solve(np.ascontiguousarray(M).ctypes.data_as(dp), v.ctypes.data_as(dp), y.ctypes.data_as(dp)) == 1
        assert np.allclose(y, np.linalg.solve(M, v), rtol=1e-9)
    Mi = -np.eye(6)
    assert lib.oracle_chol6_solve(Mi.ctypes.data_as(dp), v.ctypes.data_as(dp), y.ctypes.data_as(dp)) == 0
