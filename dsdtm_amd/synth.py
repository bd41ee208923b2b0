"""Synthetic inputs for the sparse-alignment path.

Data generation only — not part of the hot path and not an oracle. Produces what the
reference's Tracking would hand to Sprase_ImgAlign::Run: two u8 pyramids
(Frame::mvImg_Pyr, reference src/Frame.cpp:74-81), feature pixels / unit bearings /
world points (include/Feature.h:16-36, src/Frame.cpp:83-92) and the two poses.

Scene: a textured fronto-parallel plane at depth `depth` in the reference camera; the
current image is the plane-induced homography warp of the texture (bicubic).
"""
from __future__ import annotations

import dataclasses
import numpy as np

# TUM fr1 intrinsics as the reference stores them: float members (Config/kinect.yaml:50-53,63)
TUM_FR1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, f=525.0)
TUM_FR3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, f=525.0)


@dataclasses.dataclass
class Camera:
    """Pinhole intrinsics as float32 (reference include/Camera.h:138-142)."""
    fx: float
    fy: float
    cx: float
    cy: float
    f: float
    width: int
    height: int

    def __post_init__(self):
        for k in ("fx", "fy", "cx", "cy", "f"):
            setattr(self, k, float(np.float32(getattr(self, k))))

    @staticmethod
    def tum(width=640, height=480, base=TUM_FR1):
        s = width / 640.0
        return Camera(base["fx"] * s, base["fy"] * s, base["cx"] * s, base["cy"] * s,
                      base["f"] * s, width, height)

    def K(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1.0]])


def pyrdown_u8(img: np.ndarray) -> np.ndarray:
    """cv::pyrDown for CV_8UC1 restated in numpy (integer arithmetic): separable
    [1 4 6 4 1], BORDER_REFLECT_101, (sum+128)>>8, output ((w+1)/2, (h+1)/2)."""
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    a = np.pad(img.astype(np.int32), ((2, 3), (2, 3)), mode="reflect")
    # horizontal: columns 2x-2..2x+2 of the original = padded 2x .. 2x+4
    hx = (a[:, 0:2 * dw:2] + a[:, 4:2 * dw + 4:2] + 4 * (a[:, 1:2 * dw + 1:2] + a[:, 3:2 * dw + 3:2])
          + 6 * a[:, 2:2 * dw + 2:2])
    v = (hx[0:2 * dh:2] + hx[4:2 * dh + 4:2] + 4 * (hx[1:2 * dh + 1:2] + hx[3:2 * dh + 3:2])
         + 6 * hx[2:2 * dh + 2:2])
    return ((v + 128) >> 8).astype(np.uint8)


def build_pyramid(img: np.ndarray, levels: int) -> list[np.ndarray]:
    pyr = [np.ascontiguousarray(img, dtype=np.uint8)]
    for _ in range(1, levels):
        pyr.append(pyrdown_u8(pyr[-1]))
    return pyr


def make_texture(height: int, width: int, seed: int, alpha: float = 0.9) -> np.ndarray:
    """Band-limited 1/f^alpha noise, float64 in [0,255]."""
    rng = np.random.default_rng(seed)
    white = rng.standard_normal((height, width))
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.rfftfreq(width)[None, :]
    rad = np.sqrt(fx * fx + fy * fy)
    rad[0, 0] = 1.0
    spec = np.fft.rfft2(white) / rad ** alpha
    spec[0, 0] = 0.0
    # soft low-pass so level-0 gradients are meaningful at patch scale
    spec *= np.exp(-(rad / 0.25) ** 2)
    tex = np.fft.irfft2(spec, s=(height, width))
    lo, hi = np.percentile(tex, [0.5, 99.5])
    tex = np.clip((tex - lo) / (hi - lo), 0, 1) * 255.0
    return tex


def se3_exp(xi) -> np.ndarray:
    """SE(3) exponential, xi = [translation(3), rotation(3)] -> 4x4."""
    xi = np.asarray(xi, dtype=np.float64)
    ups, om = xi[:3], xi[3:]
    th = np.linalg.norm(om)
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        R = np.eye(3) + Om
        V = np.eye(3) + 0.5 * Om
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def pose_error(Ta: np.ndarray, Tb: np.ndarray) -> tuple[float, float]:
    """(rotation angle of Ra Rb^T [rad], ||ta - tb|| [m]) for 3x4/4x4 [R|t]."""
    Ta = np.asarray(Ta).reshape(-1, 4)[:3]
    Tb = np.asarray(Tb).reshape(-1, 4)[:3]
    R = Ta[:, :3] @ Tb[:, :3].T
    # robust angle: from the skew part (small angles) and trace
    s = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    c = 0.5 * (np.trace(R) - 1.0)
    ang = float(np.arctan2(np.linalg.norm(s), c))
    return ang, float(np.linalg.norm(Ta[:, 3] - Tb[:, 3]))


def warp_plane(tex: np.ndarray, cam: Camera, T_cur_ref: np.ndarray, depth: float) -> np.ndarray:
    """Current image of a fronto-parallel plane z=depth (reference camera frame)."""
    from scipy.ndimage import map_coordinates
    K = cam.K()
    R, t = T_cur_ref[:3, :3], T_cur_ref[:3, 3]
    n = np.array([0.0, 0.0, 1.0])
    Hcr = K @ (R + np.outer(t, n) / depth) @ np.linalg.inv(K)   # ref px -> cur px
    Hrc = np.linalg.inv(Hcr)
    h, w = tex.shape
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    den = Hrc[2, 0] * uu + Hrc[2, 1] * vv + Hrc[2, 2]
    xr = (Hrc[0, 0] * uu + Hrc[0, 1] * vv + Hrc[0, 2]) / den
    yr = (Hrc[1, 0] * uu + Hrc[1, 1] * vv + Hrc[1, 2]) / den
    out = map_coordinates(tex, [yr, xr], order=3, mode="reflect")
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


@dataclasses.dataclass
class AlignScene:
    cam: Camera
    ref_pyr: list
    cur_pyr: list
    px: np.ndarray        # N x 2 float32   Feature::mpx
    bearing: np.ndarray   # N x 3 float64   Feature::mNormal
    p_world: np.ndarray   # N x 3 float64   MapPoint position
    initial: np.ndarray   # N uint8         Feature::mbInitial
    T_ref_w: np.ndarray   # 3x4 float64
    T_cur_w_seed: np.ndarray  # 3x4 (Tracking seeds cur pose with the last pose, src/Tracking.cpp:201)
    T_cur_w_true: np.ndarray  # 3x4 ground truth
    depth: float


def bearing_from_px(cam: Camera, px: np.ndarray) -> np.ndarray:
    """Frame::Add_Feature (src/Frame.cpp:83-92): Pixel2Camera(cv::Point2f, 1.0f) evaluated
    in float (src/Camera.cpp:173-178), stored to Vector3d, normalised in double."""
    px = px.astype(np.float32)
    x = (np.float32(1.0) * (px[:, 0] - np.float32(cam.cx))) / np.float32(cam.fx)
    y = (np.float32(1.0) * (px[:, 1] - np.float32(cam.cy))) / np.float32(cam.fy)
    b = np.stack([x.astype(np.float64), y.astype(np.float64), np.ones(len(px))], axis=1)
    return b / np.linalg.norm(b, axis=1, keepdims=True)


def make_scene(width=640, height=480, levels=4, n_patches=300, seed=0xD5D7,
               xi=(0.01, -0.006, 0.004, 0.004, -0.003, 0.005), depth=2.0,
               T_ref_w=None, margin=30, cam: Camera | None = None,
               frac_uninitial=0.0) -> AlignScene:
    """BASELINE config 2 by default (640x480, 4 levels, 300 patches, plane at 2 m)."""
    cam = cam or Camera.tum(width, height)
    rng = np.random.default_rng(seed ^ 0x5EED)
    tex = make_texture(height, width, seed)
    T_cr = se3_exp(xi)
    ref = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    cur = warp_plane(tex, cam, T_cr, depth)
    px = np.stack([rng.uniform(margin, width - margin, n_patches),
                   rng.uniform(margin, height - margin, n_patches)], axis=1).astype(np.float32)
    bearing = bearing_from_px(cam, px)
    X_r = bearing * (depth / bearing[:, 2:3])            # on the plane z = depth
    if T_ref_w is None:
        T_ref_w = np.eye(4)[:3]
    T_ref_w = np.asarray(T_ref_w, dtype=np.float64).reshape(-1, 4)[:3]
    Rr, tr = T_ref_w[:, :3], T_ref_w[:, 3]
    p_world = (X_r - tr) @ Rr                              # R^T (X - t)
    initial = np.ones(n_patches, dtype=np.uint8)
    if frac_uninitial > 0:
        initial[rng.random(n_patches) < frac_uninitial] = 0
    T4 = np.eye(4)
    T4[:3] = T_ref_w
    T_true = (T_cr @ T4)[:3]
    return AlignScene(cam, build_pyramid(ref, levels), build_pyramid(cur, levels), px, bearing,
                      p_world, initial, T_ref_w.copy(), T_ref_w.copy(), T_true, depth)


def make_sequence(n_frames=9, width=640, height=480, levels=4, n_patches=300, seed=7, depth=2.0, margin=30,
                  cam: Camera | None = None, t_max=0.02, w_max=0.01) -> list[AlignScene]:
    """A chained sequence: a camera moves in small steps in front of a textured plane (z = depth in the frame of
    frame 0, the world). Frame k is `cur` of pair k - 1 and `ref` of pair k — the list holds the n_frames - 1 pairs
    as AlignScenes whose ref_pyr of pair k IS (the same object as) cur_pyr of pair k - 1. Features of a reference
    frame: random pixels, unit bearings, map points where the rays meet the plane; reference poses are the true
    ones, every current frame is seeded with its reference's pose (src/Tracking.cpp:201)."""
    cam = cam or Camera.tum(width, height)
    rng = np.random.default_rng(seed ^ 0x5E9)
    tex = make_texture(height, width, seed)
    T = [np.eye(4)]
    for _ in range(n_frames - 1):
        T.append(se3_exp(random_xi(rng, t_max, w_max)) @ T[-1])
    pyrs = [build_pyramid(np.clip(np.rint(tex), 0, 255).astype(np.uint8) if k == 0 else warp_plane(tex, cam, T[k], depth), levels)
            for k in range(n_frames)]
    n = np.array([0.0, 0.0, 1.0])
    out = []
    for k in range(n_frames - 1):
        px = np.stack([rng.uniform(margin, width - margin, n_patches),
                       rng.uniform(margin, height - margin, n_patches)], axis=1).astype(np.float32)
        bearing = bearing_from_px(cam, px)
        R, t = T[k][:3, :3], T[k][:3, 3]
        C = -R.T @ t                                           # camera centre in the world
        d = bearing @ R                                        # rays in the world: R^T b
        sdist = (depth - n @ C) / (d @ n)
        p_world = C + d * sdist[:, None]
        out.append(AlignScene(cam, pyrs[k], pyrs[k + 1], px, bearing, p_world, np.ones(n_patches, dtype=np.uint8),
                              T[k][:3].copy(), T[k][:3].copy(), T[k + 1][:3].copy(), depth))
    return out


def random_xi(rng, t_max=0.02, w_max=0.01):
    return np.concatenate([rng.uniform(-t_max, t_max, 3), rng.uniform(-w_max, w_max, 3)])


def random_pose(rng, t_max=1.0, w_max=0.5):
    return se3_exp(np.concatenate([rng.uniform(-t_max, t_max, 3), rng.uniform(-w_max, w_max, 3)]))[:3]


class PoseProblem:
    """Inputs of Optimizer::PoseOptimization (src/Optimizer.cpp:20-101) for one frame."""

    def __init__(self, bearing, p_world, level, use, T_seed, T_true):
        self.bearing, self.p_world, self.level, self.use = bearing, p_world, level, use
        self.T_seed, self.T_true = T_seed, T_true


def make_pose_problem(seed=0, n=200, cam: Camera | None = None, noise_px=0.3, outlier_frac=0.05,
                      seed_t=0.03, seed_w=0.02, unused_frac=0.1, max_level=3) -> PoseProblem:
    """Map points in front of a camera at a random pose, observed with pixel noise on pyramid levels
    0..max_level (noise grows with the level, as the detector's does); a fraction of gross outliers
    (what PoseOptimization's Cauchy loss and EraseFound walk are for); a fraction of features without
    a usable map point; the seed pose is the true pose perturbed (the output of sparse alignment)."""
    cam = cam or Camera.tum()
    rng = np.random.default_rng(seed)
    T_true = random_pose(rng, 0.5, 0.3)                          # world -> camera
    level = rng.integers(0, max_level + 1, n).astype(np.int32)
    px = np.stack([rng.uniform(20, cam.width - 20, n), rng.uniform(20, cam.height - 20, n)], 1)
    depth = rng.uniform(0.8, 6.0, n)
    ray = np.stack([(px[:, 0] - cam.cx) / cam.fx, (px[:, 1] - cam.cy) / cam.fy, np.ones(n)], 1)
    p_cam = ray * depth[:, None]
    R, t = T_true[:3, :3], T_true[:3, 3]
    p_world = (p_cam - t) @ R                                     # R^T (p - t)
    obs = px + rng.standard_normal((n, 2)) * noise_px * (1 << level)[:, None]
    out = rng.random(n) < outlier_frac
    obs[out] += rng.uniform(-60, 60, (int(out.sum()), 2))
    bearing = bearing_from_px(cam, obs)
    use = (rng.random(n) >= unused_frac).astype(np.uint8)
    xi = np.concatenate([rng.uniform(-seed_t, seed_t, 3), rng.uniform(-seed_w, seed_w, 3)])
    T_seed = (se3_exp(xi) @ np.vstack([T_true[:3], [0, 0, 0, 1]]))[:3]
    return PoseProblem(np.ascontiguousarray(bearing), np.ascontiguousarray(p_world), level, use,
                       np.ascontiguousarray(T_seed), np.ascontiguousarray(T_true[:3]))
