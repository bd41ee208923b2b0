"""TUM RGB-D dataset format — the data side of the reference's drivers ("callers and data formats either side
of the path"). Test/test_Tracking.cpp:56-82 reads `associations.txt`, loads `rgb/*.png` as grayscale and `depth/*.png`
unchanged (16 bit), and Tracking converts the depth with 1 / Camera.depth_scale (src/Tracking.cpp:23,56). This module is
that reader without OpenCV — a small PNG decoder (zlib only), the grayscale conversion of cv::imread(.., GRAYSCALE), the
depth scaling, Frame::Get_FeatureDetph (src/Frame.cpp:176-199) and the trajectory writer (`timestamp tx ty tz qx qy qz qw`,
System::SaveCameraTrajectory) — so that a user who HAS fr1/xyz or fr3/walking_xyz can run BASELINE configs 1 and 3 on it
(tools/run_tum.py). Host-side I/O only: no image work happens here that the GPU path is responsible for.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np


# ---- associations.txt (Test/test_Tracking.cpp:14-43 LoadImages) -------------------------------------------------
def load_associations(path: str):
    """[(t_rgb, rgb_file, t_depth, depth_file)] — one entry per non-empty line, fields as the reference reads them
    (timestamp, rgb path, timestamp, depth path; `#` comment lines of the TUM tools are skipped)."""
    out = []
    with open(path) as f:
        for line in f:
            s = line.strip()
            if not s or s.startswith("#"):
                continue
            tok = s.split()
            if len(tok) < 4:
                raise ValueError(f"{path}: expected 't rgb t depth', got {s!r}")
            out.append((float(tok[0]), tok[1], float(tok[2]), tok[3]))
    return out


def load_groundtruth(path: str):
    """groundtruth.txt of the TUM benchmark: rows `t tx ty tz qx qy qz qw` (camera -> world). Returns (t[n], T_wc[n,4,4])."""
    rows = []
    with open(path) as f:
        for line in f:
            s = line.strip()
            if s and not s.startswith("#"):
                rows.append([float(v) for v in s.split()[:8]])
    a = np.array(rows, np.float64).reshape(-1, 8)
    T = np.tile(np.eye(4), (len(a), 1, 1))
    for i, (t, tx, ty, tz, qx, qy, qz, qw) in enumerate(a):
        T[i, :3, :3] = quat_to_matrix(qx, qy, qz, qw)
        T[i, :3, 3] = (tx, ty, tz)
    return a[:, 0].copy(), T


def quat_to_matrix(qx, qy, qz, qw):
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    return np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                     [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                     [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])


def matrix_to_quat(R):
    """(qx, qy, qz, qw), qw >= 0 — Eigen::Quaterniond(R) as SaveCameraTrajectory writes it."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = ((R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s)
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = [0.0, 0.0, 0.0]
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        q = (v[0], v[1], v[2], (R[k, j] - R[j, k]) / s)
    q = np.array(q)
    return q if q[3] >= 0 else -q


def nearest_pose(t_gt, T_gt, t, max_dt=0.02):
    """Ground-truth pose closest in time to `t` (the TUM tools' associate.py rule: |dt| <= max_dt), or None."""
    i = int(np.argmin(np.abs(t_gt - t)))
    return T_gt[i] if abs(t_gt[i] - t) <= max_dt else None


# ---- PNG (8-bit gray / RGB / RGBA, 16-bit gray; non-interlaced) -----------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def read_png(path: str) -> np.ndarray:
    """Decodes the PNG flavours the TUM benchmark uses: rgb/*.png (8-bit RGB) and depth/*.png (16-bit gray). Returns an
    (H, W) or (H, W, C) array of uint8 / uint16 — what cv::imread(.., CV_LOAD_IMAGE_UNCHANGED) hands back (channels in R, G, B
    order here)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (8, 16) or interlace != 0:
        raise ValueError(f"{path}: unsupported PNG (colour type {ctype}, bit depth {depth}, interlace {interlace})")
    bpp = channels * depth // 8
    stride = w * bpp
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    rows = np.frombuffer(raw, np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 1:                                      # Sub: a running sum per byte lane
            cur = line.copy()
            for i in range(bpp, stride, bpp):
                cur[i:i + bpp] = (cur[i:i + bpp] + cur[i - bpp:i]) & 255
        elif ft == 2:                                      # Up
            cur = (line + prev) & 255
        elif ft in (3, 4):                                 # Average / Paeth: byte-serial by definition
            cur = np.zeros(stride, np.int32)
            lp, pv = line.tolist(), prev.tolist()
            c = [0] * stride
            for i in range(stride):
                a = c[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ft == 3:
                    c[i] = (lp[i] + ((a + b) >> 1)) & 255
                else:
                    d = pv[i - bpp] if i >= bpp else 0
                    p = a + b - d
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - d)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else d)
                    c[i] = (lp[i] + pr) & 255
            cur = np.array(c, np.int32)
        else:
            raise ValueError(f"{path}: bad filter type {ft}")
        out[y] = cur
        prev = cur
    if depth == 16:
        img = out.reshape(h, w, channels, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]             # big-endian samples
    else:
        img = out.reshape(h, w, channels)
    return img[:, :, 0] if channels == 1 else img


def write_png(path: str, img: np.ndarray):
    """Writes an 8-bit gray / RGB or 16-bit gray image as a PNG (filter 0 on every row): the test-data side of read_png."""
    img = np.asarray(img)
    if img.dtype == np.uint16:
        assert img.ndim == 2
        ctype, depth, body = 0, 16, img.astype(">u2").tobytes()
        stride = img.shape[1] * 2
    else:
        assert img.dtype == np.uint8 and (img.ndim == 2 or img.shape[2] == 3)
        ctype, depth = (0, 8) if img.ndim == 2 else (2, 8)
        body = np.ascontiguousarray(img).tobytes()
        stride = img.shape[1] * (1 if img.ndim == 2 else 3)
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + body[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(typ, payload):
        return struct.pack(">I", len(payload)) + typ + payload + struct.pack(">I", zlib.crc32(typ + payload) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def to_gray(img: np.ndarray) -> np.ndarray:
    """cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE) of an 8-bit colour PNG (Test/test_Tracking.cpp:75): OpenCV 2.4's PNG
    reader lets libpng convert, png_set_rgb_to_gray(.., 0.299, 0.587) — 15-bit fixed point,
    (9798 R + 19235 G + 3735 B + 16384) >> 15 (libpng >= 1.5). A gray PNG passes through."""
    img = np.asarray(img)
    if img.ndim == 2:
        return img.astype(np.uint8)
    if img.shape[2] < 3:                                    # gray + alpha (colour type 4): the gray channel
        return img[:, :, 0].astype(np.uint8)
    r, g, b = (img[:, :, k].astype(np.uint32) for k in range(3))
    return ((9798 * r + 19235 * g + 3735 * b + 16384) >> 15).astype(np.uint8)


def depth_to_metres(depth_u16: np.ndarray, depth_scale: float) -> np.ndarray:
    """tDImg.convertTo(tDImg, CV_32F, 1.0f / mDepthScale) (src/Tracking.cpp:56)."""
    return (depth_u16.astype(np.float32) * np.float32(1.0 / np.float32(depth_scale))).astype(np.float32)


def cv_round(x: float) -> int:
    return int(np.rint(x))                                  # cvRound: round half to even (SSE2 cvtsd2si)


def get_feature_depth(depth_m: np.ndarray, px) -> float:
    """Frame::Get_FeatureDetph (src/Frame.cpp:176-199): the depth at the rounded pixel, else the first non-zero of its
    left / upper / right / lower neighbour, else -1."""
    x, y = cv_round(float(px[0])), cv_round(float(px[1]))
    h, w = depth_m.shape
    if not (0 <= x < w and 0 <= y < h):                    # (the reference indexes the cv::Mat unchecked; a pixel outside has no depth)
        return -1.0
    d = depth_m[y, x]
    if d != 0:
        return float(d)
    for dx, dy in ((-1, 0), (0, -1), (1, 0), (0, 1)):
        xx, yy = x + dx, y + dy
        if 0 <= xx < w and 0 <= yy < h and depth_m[yy, xx] != 0:
            return float(depth_m[yy, xx])
    return -1.0


class TumSequence:
    """A dataset directory in the benchmark's layout: associations.txt, rgb/, depth/, optionally groundtruth.txt."""

    def __init__(self, root: str, depth_scale: float = 5000.0):
        self.root = root
        self.depth_scale = float(depth_scale)
        self.entries = load_associations(os.path.join(root, "associations.txt"))
        gt = os.path.join(root, "groundtruth.txt")
        self.gt = load_groundtruth(gt) if os.path.exists(gt) else None

    def __len__(self):
        return len(self.entries)

    def frame(self, i: int):
        """(timestamp, gray u8 (H, W), depth in metres f32 (H, W), ground-truth T_wc 4x4 or None)."""
        t, rgb, _, dep = self.entries[i]
        gray = to_gray(read_png(os.path.join(self.root, rgb)))
        depth = depth_to_metres(read_png(os.path.join(self.root, dep)), self.depth_scale)
        T = nearest_pose(self.gt[0], self.gt[1], t) if self.gt is not None else None
        return t, gray, depth, T


def write_trajectory(path: str, stamps, poses_c2w):
    """CameraTrajectory.txt in the benchmark's format from world->camera poses [R|t] (what the path produces): each row is
    the camera's pose in the world, `t tx ty tz qx qy qz qw`."""
    with open(path, "w") as f:
        for t, T in zip(stamps, poses_c2w):
            T = np.asarray(T, np.float64).reshape(-1, 4)[:3]
            R, tr = T[:, :3], T[:, 3]
            Rw, tw = R.T, -R.T @ tr
            q = matrix_to_quat(Rw)
            f.write(f"{t:.6f} {tw[0]:.7f} {tw[1]:.7f} {tw[2]:.7f} {q[0]:.7f} {q[1]:.7f} {q[2]:.7f} {q[3]:.7f}\n")
