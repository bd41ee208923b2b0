"""Independent numpy (float64) restatement of the reference path, written from the reference
text separately from oracle/dsdtm_oracle.c (rotation matrices + numpy solve instead of
quaternions + LDLT; vectorised over patches). Used only to cross-check the C oracle: two
independent readings of src/Sprase_ImageAlign.cpp and src/Feature_alignment.cpp must agree.
"""
from __future__ import annotations

import numpy as np


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(x):
    ups, om = x[:3], x[3:]
    th = np.linalg.norm(om)
    Om = hat(om)
    if th < 1e-10:
        R = np.eye(3) + Om
        V = R
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def jacobian_ba(X):
    """GetJocabianBA, src/Sprase_ImageAlign.cpp:169-193, vectorised: (n,2,6)."""
    x, y, zi = X[:, 0], X[:, 1], 1.0 / X[:, 2]
    zi2 = zi * zi
    J = np.zeros((len(X), 2, 6))
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


def bilinear_window(img, fu, fv, su, sv, offs_r, offs_c):
    """value at (fv+r+sv, fu+c+su) for all r in offs_r, c in offs_c; img float64; vectorised over n."""
    r = fv[:, None, None] + offs_r[None, :, None]
    c = fu[:, None, None] + offs_c[None, None, :]
    w00 = ((1 - su) * (1 - sv))[:, None, None]
    w01 = (su * (1 - sv))[:, None, None]
    w10 = ((1 - su) * sv)[:, None, None]
    w11 = (su * sv)[:, None, None]
    return w00 * img[r, c] + w01 * img[r, c + 1] + w10 * img[r + 1, c] + w11 * img[r + 1, c + 1]


def sparse_align(scene, max_level, min_level, max_iters, min_fts=15, T_seed=None):
    """Sprase_ImgAlign::Run (src/Sprase_ImageAlign.cpp:29-60). Returns (T_cur_w 3x4, n_tracked, iters per level)."""
    cam = scene.cam
    n_feat = len(scene.px)
    T_cw = np.eye(4); T_cw[:3] = scene.T_cur_w_seed if T_seed is None else T_seed
    T_rw = np.eye(4); T_rw[:3] = scene.T_ref_w
    iters = [0] * 8
    if n_feat < min_fts:
        return T_cw[:3], 0, iters
    T = T_cw @ np.linalg.inv(T_rw)
    Cref = -T_rw[:3, :3].T @ T_rw[:3, 3]
    n_pts = 0
    offs = np.arange(-2, 2)
    for lvl in range(max_level - 1, min_level - 1, -1):
        ref = scene.ref_pyr[lvl].astype(np.float64)
        cur = scene.cur_pyr[lvl].astype(np.float64)
        rows, cols = ref.shape
        scale = float(np.float32(1.0 / (1 << lvl)))
        p = scene.px.astype(np.float64) * scale
        zero = np.all(scene.p_world == 0, axis=1)
        ok = (scene.initial != 0) & ~zero & (p[:, 0] - 3 >= 0) & (p[:, 1] - 3 >= 0) & \
             (p[:, 0] + 3 < cols) & (p[:, 1] + 3 < rows)
        p = p[ok]
        depth = np.linalg.norm(scene.p_world[ok] - Cref, axis=1)
        X = scene.bearing[ok] * depth[:, None]
        fu, fv = np.floor(p[:, 0]).astype(int), np.floor(p[:, 1]).astype(int)
        su, sv = p[:, 0] - fu, p[:, 1] - fv
        refp = bilinear_window(ref, fu, fv, su, sv, offs, offs)
        dx = 0.5 * (bilinear_window(ref, fu, fv, su, sv, offs, offs + 1) - bilinear_window(ref, fu, fv, su, sv, offs, offs - 1))
        dy = 0.5 * (bilinear_window(ref, fu, fv, su, sv, offs + 1, offs) - bilinear_window(ref, fu, fv, su, sv, offs - 1, offs))
        Jt = jacobian_ba(X)
        fs = float(np.float32(cam.f)) * scale
        # J rows: (n,16,6)
        J = (dx.reshape(-1, 16, 1) * Jt[:, None, 0, :] + dy.reshape(-1, 16, 1) * Jt[:, None, 1, :]) * fs
        T_old = T.copy()
        chi2 = 0.0
        for it in range(max_iters):
            Pc = X @ T[:3, :3].T + T[:3, 3]
            u = (cam.fx * Pc[:, 0] / Pc[:, 2] + cam.cx) * scale
            v = (cam.fy * Pc[:, 1] / Pc[:, 2] + cam.cy) * scale
            with np.errstate(invalid="ignore"):
                ui, vi = np.floor(u), np.floor(v)
                vis = (ui - 3 >= 0) & (vi - 3 >= 0) & (ui + 3 < cols) & (vi + 3 < rows)
            n_pts = int(vis.sum())
            iters[lvl] += 1
            if n_pts == 0:
                chi2_new = np.nan
                x = np.zeros(6)          # Eigen LDLT of a zero matrix: pseudo-inverse -> 0
            else:
                uiv, viv = ui[vis].astype(int), vi[vis].astype(int)
                curp = bilinear_window(cur, uiv, viv, u[vis] - uiv, v[vis] - viv, offs, offs)
                res = (curp - refp[vis]).reshape(-1, 16)
                Jv = J[vis]
                H = np.einsum("npi,npj->ij", Jv, Jv)
                b = np.einsum("npi,np->i", Jv, res)
                chi2_new = float((res ** 2).sum() / res.size)
                x = np.linalg.solve(H, b)
            if np.isnan(x[0]) or (it > 0 and chi2_new > chi2):
                T = T_old
                break
            T_old = T.copy()
            T = T @ se3_exp(x)
            chi2 = chi2_new
            if np.abs(x).max() <= 1e-8:
                break
    return (T @ T_rw)[:3], n_pts, iters


def align2d(img, border, patch, max_iters, px):
    """Align2DGaussNewton (src/Feature_alignment.cpp:318-417) in float32 with sequential sums."""
    f32 = np.float32
    b = border.astype(np.int32).reshape(10, 10)
    dx = (f32(0.5) * (b[1:9, 2:10] - b[1:9, 0:8]).astype(f32)).reshape(64)
    dy = (f32(0.5) * (b[2:10, 1:9] - b[0:8, 1:9]).astype(f32)).reshape(64)
    H = np.zeros((3, 3), f32)
    for i in range(64):
        J = np.array([dx[i], dy[i], 1.0], f32)
        H += np.outer(J, J).astype(f32)
    with np.errstate(all="ignore"):
        Hinv = np.linalg.inv(H.astype(np.float64)).astype(f32) if np.linalg.matrix_rank(H) == 3 else np.full((3, 3), np.nan, f32)
    u, v, mean = f32(px[0]), f32(px[1]), f32(0)
    ref = patch.astype(f32).reshape(64)
    h, w = img.shape
    flat = np.concatenate([img.reshape(-1).astype(f32), np.zeros(w + 2, f32)])
    conv = False
    for _ in range(max_iters):
        if np.isnan(u) or np.isnan(v):
            break
        ur, vr = int(np.floor(u)), int(np.floor(v))
        if ur < 4 or vr < 4 or ur > w - 4 or vr > h - 4:
            break
        sx, sy = f32(u - f32(ur)), f32(v - f32(vr))
        wTL = f32((1.0 - float(sx)) * (1.0 - float(sy)))
        wTR = f32(sx * f32(f32(1) - sy))
        wBL = f32((1.0 - float(sx)) * float(sy))
        wBR = f32(sx * sy)
        Jres = np.zeros(3, f32)
        q = 0
        for j in range(8):
            o = (vr + j - 4) * w + ur - 4
            for k in range(8):
                s = f32(f32(f32(wTL * flat[o]) + f32(wTR * flat[o + 1])) + f32(wBL * flat[o + w])) + f32(wBR * flat[o + w + 1])
                r = f32(f32(s - ref[q]) + mean)
                Jres[0] -= f32(r * dx[q]); Jres[1] -= f32(r * dy[q]); Jres[2] -= r
                o += 1; q += 1
        upd = (Hinv @ Jres).astype(f32)
        u = f32(u + upd[0]); v = f32(v + upd[1]); mean = f32(mean + upd[2])
        if f32(upd[0] * upd[0] + upd[1] * upd[1]) < f32(0.03 * 0.03):
            conv = True
            break
    return conv, np.array([u, v], np.float64)
