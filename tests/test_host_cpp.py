"""The C++ host layer (dsdtm_amd/host/dsdtm_host.hpp: reference class names over the C ABI) builds
with g++ against libdsdtm_amd.so and, on a GPU, reproduces the oracle through a driver shaped like
the reference's Test/test_SpraseImg_alignment.cpp and Test/test_Feature_alignment.cpp."""
import os
import struct
import subprocess

import numpy as np
import pytest

from dsdtm_amd import capi, synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dsdtm_amd", "host")
EXE = os.path.join(HOST, "example_align")


def build_example():
    src = os.path.join(HOST, "example_align.cpp")
    hdr = os.path.join(HOST, "dsdtm_host.hpp")
    lib = capi.lib_path()
    if (not os.path.exists(EXE)) or any(os.path.getmtime(p) > os.path.getmtime(EXE) for p in (src, hdr, lib)):
        subprocess.run(["g++", "-O2", "-std=c++14", "-Wall", "-o", EXE, src, lib,
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"],
                       check=True)
    return EXE


def dump_scene(path, sc, params, min_fts, border, patch, px0):
    with open(path, "wb") as f:
        L = len(sc.ref_pyr)
        f.write(struct.pack("<8i", L, len(sc.px), params[0], params[1], params[2], min_fts, sc.cam.width, sc.cam.height))
        f.write(struct.pack("<5f", sc.cam.fx, sc.cam.fy, sc.cam.cx, sc.cam.cy, sc.cam.f))
        for pyr in (sc.ref_pyr, sc.cur_pyr):
            for l in pyr:
                f.write(np.ascontiguousarray(l).tobytes())
        for i in range(len(sc.px)):
            f.write(sc.px[i].astype("<f4").tobytes() + sc.bearing[i].astype("<f8").tobytes() +
                    sc.p_world[i].astype("<f8").tobytes() + bytes([int(sc.initial[i])]))
        f.write(np.ascontiguousarray(sc.T_ref_w, "<f8").tobytes() + np.ascontiguousarray(sc.T_cur_w_seed, "<f8").tobytes())
        f.write(bytes(border) + bytes(patch) + np.asarray(px0, "<f8").tobytes())


def test_cpp_host_layer_builds_against_the_c_abi():
    assert os.path.exists(build_example())


@pytest.mark.gpu
def test_cpp_driver_matches_oracle(tmp_path, oracle):
    exe = build_example()
    sc = synth.make_scene(width=320, height=240, levels=3, n_patches=140, seed=77, margin=12, frac_uninitial=0.05)
    img = sc.cur_pyr[0]
    pb, p = H.make_border_patches(img, [(150.3, 101.6)])
    px0 = np.array([150.3 + 0.9, 101.6 - 0.7])
    scene = tmp_path / "scene.bin"
    dump_scene(scene, sc, (3, 0, 10), 15, pb[0], p[0], px0)
    out = subprocess.run([exe, str(scene)], capture_output=True, text=True, check=True).stdout.split("\n")
    n = int(out[0].split()[1])
    T = np.array([float(v) for v in out[1].split()[1:]]).reshape(3, 4)
    iters = [int(v) for v in out[2].split()[1:]]
    a2d = out[3].split()
    To, no, so = oracle.sparse_align(sc, 3, 0, 10)
    H.assert_pose_close(T, To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what="c++ driver")
    assert n == no and iters == so["iters"][:3]
    oko, pxo = oracle.align2d(img, pb[0], p[0], 10, px0)
    assert bool(int(a2d[1])) == oko
    assert np.allclose([float(a2d[2]), float(a2d[3])], pxo, atol=2e-3)
    # the second Run of the driver used device-resident frames whose pyramids were built on the device
    # from level 0: the scene's pyramids are pyrDown chains, pyrDown is bit-exact -> the same pose, bit for bit
    assert int(out[4].split()[1]) == n
    Tres = np.array([float(v) for v in out[5].split()[1:]]).reshape(3, 4)
    assert np.array_equal(Tres, T)
    # Feature_detector::detect of the C++ layer on the (feature-less) current frame vs the sequential restatement
    from tests import detector_restatement as R
    import math
    cell = 25
    cols, rows = math.ceil(320 / cell), math.ceil(240 / cell)
    cells = oracle.detect_cells(sc.cur_pyr, 3, cell, cols, rows, None, 5.0)
    want_det = R.detect(cells, 320, 240, cell, 200, [], [], 30)
    det = [int(v) for v in out[6].split()[1:]]
    assert det[0] == len(want_det) and det[0] > 10
    assert [tuple(det[1 + 3 * i:4 + 3 * i]) for i in range(det[0])] == want_det
