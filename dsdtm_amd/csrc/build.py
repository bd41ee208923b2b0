"""Builds libdsdtm_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dsdtm_amd/csrc/build.py [--grid float|double] [--force]
"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.cpp", "sparse_align.hip", "align2d.hip", "pyrdown.hip", "warp.hip", "detect.hip", "pose_opt.hip",
           "selftest.hip"]
HEADERS = ["kernels.h", "device_math.h", os.path.join("..", "..", "include", "dsdtm_amd.h")]
OUT = os.path.join(HERE, "libdsdtm_amd.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(grid="double", force=False, verbose=True, extra=()):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-DSA_GRID_T=" + grid, *extra, "-x", "hip"]
    cmd += [os.path.join(HERE, s) for s in SOURCES]
    cmd += ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default=os.environ.get("DSDTM_GRID_T", "double"), choices=["double", "float"])
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D for experiments, e.g. SA_WAVES_PER_EU=5")
    a = ap.parse_args()
    build(a.grid, a.force, extra=["-D" + d for d in a.define])
    print(OUT)
