"""The TUM RGB-D dataset format (dsdtm_amd/tum.py; the reference's drivers read it: Test/test_Tracking.cpp:56-82) on a
miniature dataset written by the test in the same file formats, and tools/run_tum.py on it (GPU)."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from dsdtm_amd import synth, tum

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _png_with_filters(path, img16_or_8, filters):
    """A PNG whose rows use the given filter types in turn (1 Sub, 2 Up, 3 Average, 4 Paeth): what real encoders emit."""
    import struct
    img = np.asarray(img16_or_8)
    if img.dtype == np.uint16:
        raw = img.astype(">u2").view(np.uint8).reshape(img.shape[0], -1)
        ctype, depth, bpp = 0, 16, 2
    elif img.ndim == 3:
        raw, ctype, depth, bpp = img.reshape(img.shape[0], -1), 2, 8, 3
    else:
        raw, ctype, depth, bpp = img, 0, 8, 1
    h, stride = raw.shape
    out = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = raw[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - a
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pr = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            f = cur - pr
        out += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(typ, payload):
        return struct.pack(">I", len(payload)) + typ + payload + struct.pack(">I", zlib.crc32(typ + payload) & 0xffffffff)
    w = img.shape[1]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(out), 9)[:7]) + chunk(b"IDAT", zlib.compress(bytes(out), 9)[7:]) + chunk(b"IEND", b""))


def test_png_round_trip_all_filters(tmp_path):
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    d16 = rng.integers(0, 65536, (37, 53), dtype=np.uint16)
    for name, img in (("rgb", rgb), ("gray", gray), ("d16", d16)):
        tum.write_png(str(tmp_path / f"{name}.png"), img)
        assert np.array_equal(tum.read_png(str(tmp_path / f"{name}.png")), img)
        _png_with_filters(str(tmp_path / f"{name}_f.png"), img, [1, 2, 3, 4, 0])       # two IDAT chunks, every filter type
        assert np.array_equal(tum.read_png(str(tmp_path / f"{name}_f.png")), img)
    with pytest.raises(ValueError):
        (tmp_path / "x.png").write_bytes(b"not a png")
        tum.read_png(str(tmp_path / "x.png"))


def test_gray_conversion_depth_scale_and_feature_depth():
    rgb = np.zeros((2, 3, 3), np.uint8)
    rgb[0, 0] = (255, 255, 255); rgb[0, 1] = (255, 0, 0); rgb[0, 2] = (0, 255, 0); rgb[1, 0] = (0, 0, 255); rgb[1, 1] = (12, 200, 99)
    g = tum.to_gray(rgb)
    assert g[0, 0] == 255 and g[0, 1] == (9798 * 255 + 16384) >> 15 == 76 and g[0, 2] == 150 and g[1, 0] == 29
    assert g[1, 1] == (9798 * 12 + 19235 * 200 + 3735 * 99 + 16384) >> 15
    d = tum.depth_to_metres(np.array([[0, 5000], [10000, 1]], np.uint16), 5000.0)
    assert d.dtype == np.float32 and d[0, 1] == np.float32(5000) * np.float32(1.0 / np.float32(5000.0)) and d[0, 0] == 0
    depth = np.zeros((5, 5), np.float32)
    depth[2, 1] = 1.5; depth[1, 2] = 2.5                     # left and upper neighbour of (2, 2): the left one is tried first
    assert tum.get_feature_depth(depth, (2.2, 1.8)) == 1.5   # cvRound -> (2, 2), zero there -> dx/dy order of src/Frame.cpp:185-186
    assert tum.get_feature_depth(depth, (0.6, 2.4)) == 1.5   # cvRound(0.6) = 1, cvRound(2.4) = 2
    assert tum.get_feature_depth(depth, (3.5, 3.5)) == -1.0  # half to even: (4, 4); nothing around
    assert tum.cv_round(2.5) == 2 and tum.cv_round(3.5) == 4 and tum.cv_round(-0.5) == 0


def _write_dataset(root, n_frames=4, W=320, Hh=240, depth_m=2.0):
    """A plane in front of a moving camera as rgb/ depth/ associations.txt groundtruth.txt. Returns (cam, world->camera poses)."""
    os.makedirs(os.path.join(root, "rgb")); os.makedirs(os.path.join(root, "depth"))
    cam = synth.Camera.tum(W, Hh)
    tex = synth.make_texture(Hh, W, 0x70)
    rng = np.random.default_rng(5)
    xi = np.zeros(6)
    poses, lines, gt = [], [], ["# ground truth trajectory", "# timestamp tx ty tz qx qy qz qw"]
    for k in range(n_frames):
        if k:
            xi = xi + np.concatenate([rng.uniform(-0.01, 0.01, 3), rng.uniform(-0.005, 0.005, 3)])
        T_cr = synth.se3_exp(xi)                            # camera k <- reference (= world)
        gray = np.clip(np.rint(tex), 0, 255).astype(np.uint8) if k == 0 else synth.warp_plane(tex, cam, T_cr, depth_m)
        # the plane z = depth_m of the world seen from camera k: z-depth per pixel
        K = cam.K()
        R, t = T_cr[:3, :3], T_cr[:3, 3]
        uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(Hh, dtype=np.float64))
        rays = np.stack([(uu - cam.cx) / cam.fx, (vv - cam.cy) / cam.fy, np.ones_like(uu)], -1)      # camera frame, z = 1
        n_c = R @ np.array([0.0, 0.0, 1.0])                 # plane normal in the camera frame; plane: n_w . X_w = depth_m
        d_c = depth_m + n_c @ t                              # n_c . X_c = depth_m + n_c . t
        z = d_c / (rays @ n_c)
        t_s = 1305031102.0 + 0.033 * k
        rgb = np.stack([gray, gray, gray], -1)               # R = G = B: the gray conversion returns the value itself (+-0)
        tum.write_png(os.path.join(root, "rgb", f"{t_s:.6f}.png"), rgb)
        tum.write_png(os.path.join(root, "depth", f"{t_s:.6f}.png"), np.clip(np.rint(z * 5000.0), 0, 65535).astype(np.uint16))
        lines.append(f"{t_s:.6f} rgb/{t_s:.6f}.png {t_s:.6f} depth/{t_s:.6f}.png")
        T_wc = np.linalg.inv(T_cr)
        q = tum.matrix_to_quat(T_wc[:3, :3])
        gt.append(f"{t_s + 0.001:.4f} {T_wc[0, 3]:.6f} {T_wc[1, 3]:.6f} {T_wc[2, 3]:.6f} {q[0]:.8f} {q[1]:.8f} {q[2]:.8f} {q[3]:.8f}")
        poses.append(T_cr[:3].copy())
    with open(os.path.join(root, "associations.txt"), "w") as f:
        f.write("\n".join(lines) + "\n\n")                   # a trailing empty line, as the benchmark's files have
    with open(os.path.join(root, "groundtruth.txt"), "w") as f:
        f.write("\n".join(gt) + "\n")
    return cam, poses


def test_sequence_reader_and_trajectory_writer(tmp_path):
    root = str(tmp_path / "mini")
    cam, poses = _write_dataset(root, n_frames=3)
    seq = tum.TumSequence(root)
    assert len(seq) == 3
    t, gray, depth, T_wc = seq.frame(2)
    assert gray.shape == (240, 320) and gray.dtype == np.uint8 and depth.dtype == np.float32
    assert abs(float(np.median(depth)) - 2.0) < 0.05
    # the gray of an R = G = B image is that value (the three coefficients sum to 2^15)
    assert np.array_equal(gray, tum.read_png(os.path.join(root, seq.entries[2][1]))[:, :, 0])
    assert T_wc is not None and np.allclose(np.linalg.inv(T_wc)[:3], poses[2], atol=1e-5)
    out = str(tmp_path / "traj.txt")
    tum.write_trajectory(out, [e[0] for e in seq.entries], poses)
    t_back, T_back = tum.load_groundtruth(out)
    assert np.allclose(t_back, [e[0] for e in seq.entries], atol=1e-6)
    for k in range(3):
        assert np.allclose(np.linalg.inv(T_back[k])[:3], poses[k], atol=2e-6)
    # quaternion round trip incl. the trace <= 0 branches
    for xi in ([0, 0, 0, 3.0, 0.1, 0.0], [0, 0, 0, 0.1, 3.0, 0.2], [0, 0, 0, 0.0, 0.2, 3.1], [0, 0, 0, 0.3, -0.2, 0.1]):
        R = synth.se3_exp(xi)[:3, :3]
        assert np.allclose(tum.quat_to_matrix(*tum.matrix_to_quat(R)), R, atol=1e-12)


@pytest.mark.gpu
def test_run_tum_driver_on_a_miniature_dataset(tmp_path, gpu_ctx):
    """tools/run_tum.py end to end: PNGs + associations.txt in, detector + depth lookup + Sprase_ImgAlign::Run per frame,
    CameraTrajectory.txt out; the recovered camera centres agree with groundtruth.txt to millimetres."""
    root = str(tmp_path / "mini")
    cam, poses = _write_dataset(root, n_frames=4)
    out = str(tmp_path / "CameraTrajectory.txt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_tum.py"), root, "--out", out, "--keyframe-every", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == 4 and "with depth" in lines[0]
    for l in lines[1:]:
        assert int(l.split("tracked ")[1].split()[0]) > 50
        assert float(l.split("translation_error ")[1].split()[0]) < 5e-3
    t_back, T_back = tum.load_groundtruth(out)
    assert len(t_back) == 4
    for k in range(4):
        ang, dt = synth.pose_error(np.linalg.inv(T_back[k])[:3], poses[k])
        assert ang < 2e-3 and dt < 5e-3, (k, ang, dt)


@pytest.mark.gpu
def test_cpp_rgbd_driver_on_a_converted_miniature_dataset(tmp_path, gpu_ctx):
    """The C++ driver of BASELINE config 1 (dsdtm_amd/host/example_rgbd.cpp, the shape of Test/test_SpraseImg_alignment.cpp) on
    a dataset in the TUM layout: tools/tum_to_rgbd_bin.py converts PNGs + associations + ground truth (CPU), the driver detects
    on the first frame, takes depths from its depth image (a hole punched into it: features there are dropped), aligns every
    later frame against it and prints the error against ground truth."""
    from tests.test_host_cpp import build_example
    root = str(tmp_path / "mini")
    W, Hh = 640, 480
    cam, poses = _write_dataset(root, n_frames=4, W=W, Hh=Hh)
    # a hole in the first depth image (no measurement), as real sensors leave
    seq = tum.TumSequence(root)
    dpath = os.path.join(root, seq.entries[0][3])
    d16 = tum.read_png(dpath).copy()
    d16[100:220, 200:360] = 0
    tum.write_png(dpath, d16)
    out = str(tmp_path / "seq.bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tum_to_rgbd_bin.py"), root, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = build_example("example_rgbd")
    feats = str(tmp_path / "features.bin")
    lines = subprocess.run([exe, out, feats], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    n = int(lines[0].split()[1])
    assert n >= 80, lines[0]
    raw = np.fromfile(feats, dtype=np.uint8)[4:].reshape(n, 56)
    px = raw[:, :8].copy().view("<f4").reshape(n, 2)
    assert not ((px[:, 0] >= 200) & (px[:, 0] < 360) & (px[:, 1] >= 100) & (px[:, 1] < 220)).any()     # none inside the hole
    for k in range(1, 4):
        tok = lines[k].split()
        assert tok[0] == "frame" and int(tok[1]) == k and int(tok[3]) > 60
        assert float(tok[5]) < 5e-3 and float(tok[7]) < 2e-3, lines[k]       # translation error [m], angular distance [rad] vs ground truth
