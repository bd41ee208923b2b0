"""Builds libdsdtm_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dsdtm_amd/csrc/build.py [--force] [--define X=1 ...] [--flag F ...]

Every source is compiled to its own object under csrc/build/ (only when it or a header is newer; in
parallel; objects are keyed by the flag set), then linked: a change to one kernel file recompiles that file
only. The flag set the library was linked from is recorded beside it (libdsdtm_amd.so.tag): a request for
another flag set relinks even when every object is fresh, so an experiment build (--define / --flag) can
never be mistaken for the default one.
"""
import argparse
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.cpp", "sparse_align.hip", "align2d.hip", "pyrdown.hip", "warp.hip", "match.hip", "detect.hip", "pose_opt.hip",
           "selftest.hip"]
HEADERS = ["kernels.h", "device_math.h", "warp_body.h", "align2d_body.h", os.path.join("..", "..", "include", "dsdtm_amd.h")]
OUT = os.path.join(HERE, "libdsdtm_amd.so")
TAG = OUT + ".tag"
OBJ_DIR = os.path.join(HERE, "build")


def _flags(extra):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", *extra]


def _tag(flags):
    return hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]


def _obj(src, flags):
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{_tag(flags)}.o")


def source_sha(extra=()):
    """sha256 (16 hex digits) over the library's sources, headers and compiler flags: what a profile is tied to besides
    the hash of the binary it ran with (hipcc's objects are not reproducible byte for byte, the sources are)."""
    h = hashlib.sha256(" ".join(_flags(list(extra))).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _linked_tag():
    try:
        with open(TAG) as f:
            return f.read().strip()
    except OSError:
        return None


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build(extra=()):
    flags = _flags(list(extra))
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = [_obj(s, flags) for s in SOURCES]
    return (_linked_tag() != _tag(flags) or _stale(OUT, objs)
            or any(_stale(o, [os.path.join(HERE, s)] + hdrs) for o, s in zip(objs, SOURCES)))


def build(force=False, verbose=True, extra=()):
    flags = _flags(list(extra))
    hipcc = os.environ.get("HIPCC", "hipcc")
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for s in SOURCES:
        o = _obj(s, flags)
        if force or _stale(o, [os.path.join(HERE, s)] + hdrs):
            jobs.append([hipcc, *flags, "-x", "hip", "-c", os.path.join(HERE, s), "-o", o])
    objs = [_obj(s, flags) for s in SOURCES]
    if not jobs and not _stale(OUT, objs) and not force and _linked_tag() == _tag(flags):
        return OUT

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if os.path.exists(TAG):
        os.remove(TAG)          # no tag while the library is being replaced
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", *objs, "-o", OUT])
    with open(TAG, "w") as f:
        f.write(_tag(flags) + "\n")
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D for experiments")
    ap.add_argument("--flag", action="append", default=[], help="extra raw compiler flag for experiments, e.g. --flag=-mllvm --flag=-amdgpu-sched-strategy=max-ilp")
    a = ap.parse_args()
    build(a.force, extra=["-D" + d for d in a.define] + list(a.flag))
    print(OUT)
