"""Independent numpy restatement of Optimizer::PoseOptimization (src/Optimizer.cpp:20-101) used to
cross-check oracle/pose_opt_oracle.c: rotations through scipy's Rotation (not the Sophus restatement),
the damped least-squares step through LAPACK QR (numpy.linalg.qr). Test infrastructure only.

Structure follows Ceres' TrustRegionMinimizer / LevenbergMarquardtStrategy as the oracle's header
describes; this file shares no code with it.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def _pose(x):
    return Rotation.from_rotvec(x[3:]).as_matrix(), x[:3]


def plus(x, d):
    """PoseLocalParameterization::Plus (include/Optimizer.h:222-236)."""
    Ro, to = _pose(x)
    Rd, td = _pose(d)
    Rn = Rd @ Ro
    tn = Rd @ to + td
    return np.concatenate([tn, Rotation.from_matrix(Rn).as_rotvec()])


def residuals(x, obs, pw, inv):
    R, t = _pose(x)
    p = pw @ R.T + t
    return (obs - p[:, :2] / p[:, 2:3]) / inv[:, None], p


def jacobians(p):
    """include/Optimizer.h:165-190 (not divided by 1 << level)."""
    x, y, zi = p[:, 0], p[:, 1], 1.0 / p[:, 2]
    zi2 = zi * zi
    J = np.zeros((len(p), 2, 6))
    J[:, 0, 0] = -zi
    J[:, 0, 2] = x * zi2
    J[:, 0, 3] = y * J[:, 0, 2]
    J[:, 0, 4] = -(1.0 + x * J[:, 0, 2])
    J[:, 0, 5] = y * zi
    J[:, 1, 1] = -zi
    J[:, 1, 2] = y * zi2
    J[:, 1, 3] = 1.0 + y * J[:, 1, 2]
    J[:, 1, 4] = -x * J[:, 1, 2]
    J[:, 1, 5] = -x * zi
    return J


def cost_only(x, obs, pw, inv):
    r, _ = residuals(x, obs, pw, inv)
    s = (r * r).sum(1)
    return 0.5 * np.log1p(s).sum()          # 1/2 sum rho(s), Cauchy(1): rho = log(1 + s)


def evaluate(x, obs, pw, inv):
    r, p = residuals(x, obs, pw, inv)
    s = (r * r).sum(1)
    cost = 0.5 * np.log(1.0 + s).sum()
    w = np.sqrt(1.0 / (1.0 + s))            # sqrt(rho'), corrector with alpha = 0
    J = jacobians(p) * w[:, None, None]
    rc = r * w[:, None]
    J = J.reshape(-1, 6)
    rc = rc.reshape(-1)
    return cost, rc, J, J.T @ rc


def pose_optimization(bearing, p_world, level, use, T_cur_w, max_iterations=100):
    use = np.asarray(use).astype(bool)
    b = np.asarray(bearing, np.float64)[use]
    pw = np.asarray(p_world, np.float64)[use]
    inv = (1 << np.asarray(level)[use]).astype(np.float64)
    obs = b[:, :2] / b[:, 2:3]
    T = np.asarray(T_cur_w, np.float64).reshape(3, 4)
    x = np.concatenate([T[:, 3], Rotation.from_matrix(T[:, :3]).as_rotvec()])
    trace = []
    info = dict(iterations=0, successful_steps=0, termination=6, n_residual_blocks=int(use.sum()))
    if use.sum():
        cost, r, J, g = evaluate(x, obs, pw, inv)
        info["initial_cost"] = cost
        scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
        J = J * scale
        gmax = np.abs(x - plus(x, -g)).max()
        radius, dec, reuse, invalid, it, succ = 1e4, 2.0, False, 0, 0, 0
        diag = None
        while True:
            trace.append((cost, radius, gmax, succ))
            if it >= max_iterations:
                term = 3; break
            if gmax <= 1e-10:
                term = 2; break
            if radius <= 1e-32:
                term = 4; break
            it += 1
            if not reuse:
                diag = np.clip((J * J).sum(0), 1e-6, 1e32)
            lm = np.sqrt(diag / radius)
            A = np.vstack([J, np.diag(lm)])
            rhs = np.concatenate([r, np.zeros(6)])
            Q, Rm = np.linalg.qr(A)
            step = -np.linalg.solve(Rm, Q.T @ rhs)
            reuse = True
            mr = J @ step
            model_change = -(mr @ (r + mr / 2.0))
            if not (np.isfinite(step).all() and model_change > 0):
                invalid += 1
                if invalid >= 5:
                    term = 5; break
                radius /= dec; dec *= 2
                continue
            invalid = 0
            cand = plus(x, step * scale)
            ccost = cost_only(cand, obs, pw, inv)
            if np.linalg.norm(x - cand) <= 1e-8 * (np.linalg.norm(x) + 1e-8):
                term = 1; break
            change = cost - ccost
            if abs(change) <= 1e-6 * cost:
                term = 0; break
            rho = change / model_change
            if rho > 1e-3:
                x = cand
                cost, r, J, g = evaluate(x, obs, pw, inv)
                J = J * scale
                gmax = np.abs(x - plus(x, -g)).max()
                succ += 1
                radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
                dec, reuse = 2.0, False
            else:
                radius /= dec; dec *= 2; reuse = True
        info.update(iterations=it, successful_steps=succ, termination=term, final_cost=cost)
    R, t = _pose(x)
    Tn = np.concatenate([R, t[:, None]], 1)
    rn = np.zeros(0)
    if use.sum():
        rr, _ = residuals(x, obs, pw, inv)
        rn = np.sqrt((rr * rr).sum(1))
    info["x"] = x
    return Tn, rn, info, np.array(trace)


def robust_cost(T, bearing, p_world, level, use):
    """1/2 sum log(1 + |r|^2) at pose T — for an independent minimiser (scipy BFGS) in the tests."""
    use = np.asarray(use).astype(bool)
    b = np.asarray(bearing)[use]
    T = np.asarray(T).reshape(3, 4)
    p = np.asarray(p_world)[use] @ T[:, :3].T + T[:, 3]
    r = (b[:, :2] / b[:, 2:3] - p[:, :2] / p[:, 2:3]) / (1 << np.asarray(level)[use])[:, None]
    return 0.5 * np.log1p((r * r).sum(1)).sum()
