"""Builds the HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dsdtm_amd/csrc/build.py            # the RELEASE library  libdsdtm_amd.so       (the product)
    python dsdtm_amd/csrc/build.py --diag     # the DIAGNOSTIC build libdsdtm_amd_diag.so  (tools/, diag-marked tests)
    python dsdtm_amd/csrc/build.py --all      # both
    ... [--force] [--define X=1 ...] [--flag F ...]   (extra flags: experiments, see below)

RELEASE (default): exports exactly the symbols include/dsdtm_amd.h declares (linker version script exports.map), reads no
environment variable, has no dsdtm_debug_* entry, and does not contain the kernels only a diagnostic switch selects
(align2d_kernel<true>, the 8-per-wave Align2D, warp groups of 8/32/64, match groups of 32/64, the stamps instantiation of the
register kernel, selftest.hip, the two-kernel FindMatchDirect path). bench.py, __graft_entry__.smoke() and every parity test
load this one. DIAG (-DDSDTM_DIAG): the same sources with the switches of kernels.h as process-wide variables
(DSDTM_* environment, dsdtm_debug_set_option), the dsdtm_debug_* entries and those kernels; loaded by tools/ and by the tests
marked `diag` (fault injection: a team member that stays away, the epoch wrap, A/B of superseded paths).

Every source is compiled to its own object under csrc/build/ (only when it or a header is newer; in parallel; objects are
keyed by the flag set), then linked: a change to one kernel file recompiles that file only. The flag set a library was
linked from is recorded beside it (<lib>.tag): a request for another flag set relinks even when every object is fresh, so an
experiment build (--define / --flag) can never be mistaken for the default one.
"""
import argparse
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.cpp", "sparse_align.hip", "align2d.hip", "pyrdown.hip", "warp.hip", "match.hip", "detect.hip", "pose_opt.hip", "track.hip"]
DIAG_SOURCES = ["selftest.hip"]                      # diagnostic build only
HEADERS = ["kernels.h", "device_math.h", "warp_body.h", "align2d_body.h", "match_body.h", "exports.map", os.path.join("..", "..", "include", "dsdtm_amd.h")]
OUT = os.path.join(HERE, "libdsdtm_amd.so")
OUT_DIAG = os.path.join(HERE, "libdsdtm_amd_diag.so")
TAG = OUT + ".tag"
OBJ_DIR = os.path.join(HERE, "build")
DIAG_DEFINE = "-DDSDTM_DIAG=1"


def out_path(diag=False):
    return OUT_DIAG if diag else OUT


def _sources(diag):
    return SOURCES + (DIAG_SOURCES if diag else [])


def _flags(extra, diag=False):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", *([DIAG_DEFINE] if diag else []), *extra]


def _tag(flags):
    return hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]


def _obj(src, flags):
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{_tag(flags)}.o")


def source_sha(extra=(), diag=False):
    """sha256 (16 hex digits) over the library's sources, headers and compiler flags: what a profile is tied to besides
    the hash of the binary it ran with (hipcc's objects are not reproducible byte for byte, the sources are)."""
    h = hashlib.sha256(" ".join(_flags(list(extra), diag)).encode())
    for f in sorted(_sources(diag) + HEADERS):
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _linked_tag(out):
    try:
        with open(out + ".tag") as f:
            return f.read().strip()
    except OSError:
        return None


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build(extra=(), diag=False):
    flags = _flags(list(extra), diag)
    out = out_path(diag)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    srcs = _sources(diag)
    objs = [_obj(s, flags) for s in srcs]
    return (_linked_tag(out) != _tag(flags) or _stale(out, objs)
            or any(_stale(o, [os.path.join(HERE, s)] + hdrs) for o, s in zip(objs, srcs)))


def build(force=False, verbose=True, extra=(), diag=False):
    flags = _flags(list(extra), diag)
    out = out_path(diag)
    hipcc = os.environ.get("HIPCC", "hipcc")
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources(diag)
    jobs = []
    for s in srcs:
        o = _obj(s, flags)
        if force or _stale(o, [os.path.join(HERE, s)] + hdrs):
            jobs.append([hipcc, *flags, "-x", "hip", "-c", os.path.join(HERE, s), "-o", o])
    objs = [_obj(s, flags) for s in srcs]
    if not jobs and not _stale(out, objs) and not force and _linked_tag(out) == _tag(flags):
        return out

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    tag = out + ".tag"
    if os.path.exists(tag):
        os.remove(tag)          # no tag while the library is being replaced
    # exports.map: only dsdtm_* leaves the library (release: exactly the header's symbols; diag: + dsdtm_debug_*)
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wl,--version-script=" + os.path.join(HERE, "exports.map"),
         *objs, "-o", out])
    with open(tag, "w") as f:
        f.write(_tag(flags) + "\n")
    return out


def build_all(force=False, verbose=True):
    """The release library and the diagnostic one (what __graft_entry__.build() and the test session build)."""
    return build(force, verbose), build(force, verbose, diag=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--diag", action="store_true", help="build libdsdtm_amd_diag.so (debug entries, environment switches, superseded kernels)")
    ap.add_argument("--all", action="store_true", help="build both libraries")
    ap.add_argument("--define", action="append", default=[], help="extra -D for experiments")
    ap.add_argument("--flag", action="append", default=[], help="extra raw compiler flag for experiments, e.g. --flag=-mllvm --flag=-amdgpu-sched-strategy=max-ilp")
    a = ap.parse_args()
    extra = ["-D" + d for d in a.define] + list(a.flag)
    if a.all or not a.diag:
        print(build(a.force, extra=extra))
    if a.all or a.diag:
        print(build(a.force, extra=extra, diag=True))
