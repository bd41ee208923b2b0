"""Builds libdsdtm_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dsdtm_amd/csrc/build.py [--grid float|double] [--force]

Every source is compiled to its own object under csrc/build/ (only when it or a header is newer; in
parallel), then linked: a change to one kernel file recompiles that file only.
"""
import argparse
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.cpp", "sparse_align.hip", "align2d.hip", "pyrdown.hip", "warp.hip", "detect.hip", "pose_opt.hip",
           "selftest.hip"]
HEADERS = ["kernels.h", "device_math.h", os.path.join("..", "..", "include", "dsdtm_amd.h")]
OUT = os.path.join(HERE, "libdsdtm_amd.so")
OBJ_DIR = os.path.join(HERE, "build")


def _flags(grid, extra):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DSA_GRID_T=" + grid, *extra]


def _obj(src, flags):
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{tag}.o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build(grid="double", extra=()):
    flags = _flags(grid, extra)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = [_obj(s, flags) for s in SOURCES]
    return _stale(OUT, objs) or any(_stale(o, [os.path.join(HERE, s)] + hdrs) for o, s in zip(objs, SOURCES))


def build(grid="double", force=False, verbose=True, extra=()):
    flags = _flags(grid, list(extra))
    hipcc = os.environ.get("HIPCC", "hipcc")
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for s in SOURCES:
        o = _obj(s, flags)
        if force or _stale(o, [os.path.join(HERE, s)] + hdrs):
            jobs.append([hipcc, *flags, "-x", "hip", "-c", os.path.join(HERE, s), "-o", o])
    objs = [_obj(s, flags) for s in SOURCES]
    if not jobs and not _stale(OUT, objs) and not force:
        return OUT

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", *objs, "-o", OUT])
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default=os.environ.get("DSDTM_GRID_T", "double"), choices=["double", "float"])
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D for experiments, e.g. SA_WAVES_PER_EU=5")
    ap.add_argument("--flag", action="append", default=[], help="extra raw compiler flag for experiments, e.g. --flag=-mllvm --flag=-amdgpu-sched-strategy=max-ilp")
    a = ap.parse_args()
    build(a.grid, a.force, extra=["-D" + d for d in a.define] + list(a.flag))
    print(OUT)
