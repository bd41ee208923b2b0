#!/usr/bin/env python3
"""Generates tests/golden/fast_reference.npz: images + the corners the REFERENCE's own FAST code finds.

Unlike the alignment fixtures these vectors come from the reference itself: `make -C oracle ref`
compiles the vendored Thirdparty/fast sources where they lie under /root/reference into
oracle/_ref/libfast_ref.so, and this script runs that library (through oracle/fast_ref_shim.cpp) on
  * the reference's own test image Thirdparty/fast/test/data/test1.png (752x480, 8-bit gray; its test
    prints the known answer "BENCHMARK version extracted 167 features" at barrier 75,
    Thirdparty/fast/test/test.cpp:16-54) — decoded here with zlib, no image library needed;
  * small synthetic images (noise, low contrast, a texture; one narrower than 22 columns, which takes
    the plain detector, faster_corner_10_sse.cpp:190-193).
The fixture holds data only: pixel arrays and (x, y, score, is_nonmax) rows. Needs /root/reference:

    python tests/golden/make_golden_fast.py
"""
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dsdtm_amd import synth  # noqa: E402
from tests import oracle_lib  # noqa: E402


def read_png_gray8(path):
    """8-bit grayscale, non-interlaced PNG -> (h, w) uint8."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    i, idat, w, h = 8, b"", 0, 0
    while i < len(b):
        n, = struct.unpack(">I", b[i:i + 4]); t = b[i + 4:i + 8]; data = b[i + 8:i + 8 + n]
        if t == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, interlace) == (8, 0, 0), (depth, ctype, interlace)
        elif t == b"IDAT":
            idat += data
        i += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), np.uint8)
    prev = np.zeros(w, np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(w, np.int32)
        if f == 0: cur = line
        elif f == 2: cur = (line + prev) & 255
        else:
            for x in range(w):
                a = cur[x - 1] if x else 0
                c = prev[x - 1] if x else 0
                bb = prev[x]
                if f == 1: p = a
                elif f == 3: p = (a + bb) >> 1
                else:
                    pa, pb, pc = abs(bb - c), abs(a - c), abs(a + bb - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur; prev = cur
    return out


def main():
    assert oracle_lib.fast_ref_lib() is not None, "needs /root/reference (make -C oracle ref)"
    d = {}
    test1 = read_png_gray8("/root/reference/Thirdparty/fast/test/data/test1.png")
    assert test1.shape == (480, 752)
    r75 = oracle_lib.fast10_list_reference(test1, 75)
    assert len(r75) == 167, len(r75)            # the reference test's known answer
    d["test1"] = test1; d["test1_b75"] = r75; d["test1_b20"] = oracle_lib.fast10_list_reference(test1, 20)
    rng = np.random.default_rng(5)
    imgs = {
        "noise": rng.integers(0, 256, (60, 96), dtype=np.uint8),
        "lowc": (128 + rng.integers(-30, 31, (72, 100))).astype(np.uint8),
        "tex": np.clip(np.rint(synth.make_texture(120, 160, 9)), 0, 255).astype(np.uint8),
        "narrow": rng.integers(0, 256, (40, 21), dtype=np.uint8),
    }
    for k, img in imgs.items():
        d[k] = img; d[k + "_b20"] = oracle_lib.fast10_list_reference(img, 20)
    np.savez_compressed(os.path.join(HERE, "fast_reference.npz"), **d)
    print({k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
