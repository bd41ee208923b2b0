"""ctypes loader for the CPU oracle (oracle/_build/liboracle.so) — checker only.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

from dsdtm_amd import capi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIBS = {}


class OracleSE3(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]


def _cpu_tag() -> str:
    """Tag of the host CPU (model + ISA flags): a -march=native build is only loaded on the CPU it was made on."""
    import hashlib
    model, flags = "", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not model:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = line.split(":", 1)[1].strip()
                if model and flags:
                    break
    except OSError:
        pass
    return hashlib.sha1((model + "|" + flags).encode()).hexdigest()[:10]


def build(native: bool = False):
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR] + (["native", "NATIVE_TAG=" + _cpu_tag()] if native else []), check=True)


def _declare(lib):
    capi.declare_signatures(lib, "oracle_", with_ctx=False)
    dp = C.POINTER(C.c_double)
    lib.oracle_align2d.restype = C.c_int
    lib.oracle_align2d.argtypes = [capi.u8p, C.c_int, C.c_int, C.c_int, capi.u8p, capi.u8p, C.c_int, dp]
    lib.oracle_pyrdown.restype = None
    lib.oracle_pyrdown.argtypes = [capi.u8p, C.c_int, C.c_int, C.c_int, capi.u8p, C.c_int]
    lib.oracle_se3_from_rt.argtypes = [dp, C.POINTER(OracleSE3)]
    lib.oracle_se3_to_rt.argtypes = [C.POINTER(OracleSE3), dp]
    lib.oracle_se3_exp.argtypes = [dp, C.POINTER(OracleSE3)]
    lib.oracle_se3_mul.argtypes = [C.POINTER(OracleSE3)] * 3
    lib.oracle_se3_inverse.argtypes = [C.POINTER(OracleSE3)] * 2
    lib.oracle_se3_act.argtypes = [C.POINTER(OracleSE3), dp, dp]
    lib.oracle_ldlt6_solve.argtypes = [dp, dp, dp]
    lib.oracle_jacobian_ba.argtypes = [dp, dp]
    lib.oracle_sparse_align_batch_timed.restype = C.c_double
    lib.oracle_sparse_align_batch_timed.argtypes = [C.POINTER(capi.BatchDesc), C.POINTER(capi.Camera),
                                                    C.POINTER(capi.AlignParams), C.c_int]
    ip32 = C.POINTER(C.c_int32)
    lib.oracle_fast10_list.restype = C.c_int
    lib.oracle_fast10_list.argtypes = [capi.u8p, C.c_int, C.c_int, C.c_int, C.c_int, ip32, C.c_int]
    lib.oracle_shi_tomasi.restype = C.c_float
    lib.oracle_shi_tomasi.argtypes = [capi.u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.oracle_detect_cells.restype = None
    lib.oracle_detect_cells.argtypes = [C.POINTER(capi.Pyramid), C.c_int, C.c_int, C.c_int, C.c_int, capi.u8p, C.c_double,
                                        C.c_int, C.POINTER(C.c_float), ip32, ip32, ip32]
    lib.oracle_pose_optimization.restype = C.c_int
    lib.oracle_pose_optimization.argtypes = [dp, dp, ip32, capi.u8p, C.c_int, dp, C.POINTER(capi.PoseOptParams), C.c_int,
                                             dp, C.POINTER(capi.PoseOptSummary), dp, C.c_int]
    lib.oracle_so3_log.argtypes = [dp, dp]
    lib.oracle_pose_plus.argtypes = [dp, dp, dp]
    lib.oracle_chol6_solve.restype = C.c_int
    lib.oracle_chol6_solve.argtypes = [dp, dp, dp]


def load(native: bool = False):
    key = "native" if native else "ref"
    if key in _LIBS:
        return _LIBS[key]
    name = f"liboracle_native_{_cpu_tag()}.so" if native else "liboracle.so"
    path = os.path.join(_ORACLE_DIR, "_build", name)
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("dsdtm_oracle.c", "pose_opt_oracle.c", "dsdtm_oracle.h", "mutants.h")]
    override = os.environ.get("DSDTM_ORACLE_LIB")       # the sanitizer job (tests/test_sanitizers_cpu.py) runs the oracle's own
    if override and not native:                          # CPU tests against an ASan/UBSan build of the same sources
        path = override
    elif not os.path.exists(path) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
        build(native)
    lib = C.CDLL(path)
    _declare(lib)
    _LIBS[key] = lib
    return lib


def pose_optimization(bearing, p_world, level, use, T_cur_w, max_iterations=100, linear_solver=0, trace=False):
    """oracle_pose_optimization: returns (T 3x4, residual norms in block order, summary dict[, trace rows])."""
    lib = load()
    bearing = np.ascontiguousarray(bearing, np.float64).reshape(-1, 3)
    pw = np.ascontiguousarray(p_world, np.float64).reshape(-1, 3)
    level = np.ascontiguousarray(level, np.int32)
    use = np.ascontiguousarray(use, np.uint8)
    n = len(level)
    T = np.ascontiguousarray(T_cur_w, np.float64).reshape(12).copy()
    prm = capi.PoseOptParams(max_iterations, 0)
    sm = capi.PoseOptSummary()
    rn = np.full(max(n, 1), np.nan)
    cap = max_iterations + 2
    tr = np.zeros((cap, 4))
    dp = C.POINTER(C.c_double)
    rc = lib.oracle_pose_optimization(bearing.ctypes.data_as(dp), pw.ctypes.data_as(dp),
                                      level.ctypes.data_as(C.POINTER(C.c_int32)), use.ctypes.data_as(capi.u8p), n,
                                      T.ctypes.data_as(dp), C.byref(prm), linear_solver, rn.ctypes.data_as(dp),
                                      C.byref(sm), tr.ctypes.data_as(dp) if trace else None, cap if trace else 0)
    assert rc == 0, rc
    d = sm.as_dict()
    out = (T.reshape(3, 4), rn[:d["n_residual_blocks"]].copy(), d)
    if trace:
        out += (tr[:d["iterations"] + 1].copy(),)
    return out


# ---- the mutation build (oracle/mutants.h): the quirks of SURVEY.md §8.1 "fixed" one at a time ----
def _mutant_ids():
    """{name: id} parsed from oracle/mutants.h (one source of truth for the C build and the tests)."""
    import re
    with open(os.path.join(_ORACLE_DIR, "mutants.h")) as f:
        return {m.group(1): int(m.group(2)) for m in re.finditer(r"^\s*(MUT_[A-Z0-9_]+)\s*=\s*(\d+)", f.read(), re.M)}


MUTANTS = _mutant_ids()


def load_mutants():
    if "mut" not in _LIBS:
        subprocess.run(["make", "-s", "-C", _ORACLE_DIR, "mutants"], check=True)
        lib = C.CDLL(os.path.join(_ORACLE_DIR, "_build", "liboracle_mut.so"))
        _declare(lib)
        lib.oracle_set_mutant.argtypes = [C.c_int]
        lib.oracle_get_mutant.restype = C.c_int
        _LIBS["mut"] = lib
    return _LIBS["mut"]


@contextlib.contextmanager
def mutant(name_or_id):
    """Every wrapper of this module runs on the mutation build with ONE quirk switched inside the block
    (MUT_NONE / 0: the mutation build with nothing switched — must equal the faithful library bit for bit)."""
    mid = MUTANTS[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
    lib, faithful = load_mutants(), load()
    lib.oracle_set_mutant(mid)
    _LIBS["ref"] = lib
    try:
        yield lib
    finally:
        lib.oracle_set_mutant(0)
        _LIBS["ref"] = faithful


# ---- the reference's own FAST build (oracle/_ref/libfast_ref.so, `make -C oracle ref`) ------------
def fast_ref_lib():
    """The vendored Thirdparty/fast of the reference compiled from its own sources (only where
    /root/reference exists, or where the prebuilt file travelled); None otherwise."""
    path = os.path.join(_ORACLE_DIR, "_ref", "libfast_ref.so")
    if not os.path.exists(path):
        if not os.path.isdir("/root/reference/Thirdparty/fast/src"):
            return None
        subprocess.run(["make", "-s", "-C", _ORACLE_DIR, "ref"], check=True)
    if "fast_ref" not in _LIBS:
        lib = C.CDLL(path)
        lib.fast_ref_detect.restype = C.c_int
        lib.fast_ref_detect.argtypes = [capi.u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]
        _LIBS["fast_ref"] = lib
    return _LIBS["fast_ref"]


def _fast_list(fn, img, barrier):
    img = np.asarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((w * h, 4), np.int32)
    n = fn(img.ctypes.data_as(capi.u8p), w, h, img.strides[0], barrier, out.ctypes.data_as(C.POINTER(C.c_int32)), w * h)
    return out[:n].copy()


def fast10_list(img, barrier=20):
    """(x, y, score, is_nonmax) rows in raster order — oracle restatement."""
    return _fast_list(load().oracle_fast10_list, img, barrier)


def fast10_list_reference(img, barrier=20):
    lib = fast_ref_lib()
    return None if lib is None else _fast_list(lib.fast_ref_detect, img, barrier)


def shi_tomasi(img, u, v):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return float(load().oracle_shi_tomasi(img.ctypes.data_as(capi.u8p), img.shape[1], img.shape[0], img.strides[0], u, v))


def detect_cells(img_pyr, levels, cell_size, grid_cols, grid_rows, occupied, detection_threshold, barrier=20):
    pyr, keep = capi.pyramid_struct(img_pyr)
    G = grid_cols * grid_rows
    score = np.zeros(G, np.float32)
    cx, cy, cl = np.zeros(G, np.int32), np.zeros(G, np.int32), np.zeros(G, np.int32)
    occ = np.ascontiguousarray(occupied, np.uint8) if occupied is not None else np.zeros(G, np.uint8)
    ip = C.POINTER(C.c_int32)
    load().oracle_detect_cells(C.byref(pyr), levels, cell_size, grid_cols, grid_rows, occ.ctypes.data_as(capi.u8p),
                               float(detection_threshold), barrier, score.ctypes.data_as(C.POINTER(C.c_float)),
                               cx.ctypes.data_as(ip), cy.ctypes.data_as(ip), cl.ctypes.data_as(ip))
    return score, cx, cy, cl


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def sparse_align(scene, max_level, min_level, max_iters, min_fts=15, T_seed=None, native=False):
    """Runs oracle_sparse_align on an AlignScene-like object.
    Returns (T_cur_w 3x4, n_tracked, stats dict)."""
    lib = load(native)
    ref, k1 = capi.pyramid_struct(scene.ref_pyr)
    cur, k2 = capi.pyramid_struct(scene.cur_pyr)
    cam = capi.camera_struct(scene.cam)
    px = np.ascontiguousarray(scene.px, dtype=np.float32)
    bearing = np.ascontiguousarray(scene.bearing, dtype=np.float64)
    pw = np.ascontiguousarray(scene.p_world, dtype=np.float64)
    ini = np.ascontiguousarray(scene.initial, dtype=np.uint8)
    Tr = np.ascontiguousarray(scene.T_ref_w, dtype=np.float64).reshape(12).copy()
    Tc = np.ascontiguousarray(scene.T_cur_w_seed if T_seed is None else T_seed, dtype=np.float64).reshape(12).copy()
    prm = capi.AlignParams(max_level, min_level, max_iters, min_fts)
    nt = C.c_int(0)
    st = capi.AlignStats()
    rc = lib.oracle_sparse_align(C.byref(ref), C.byref(cur), C.byref(cam),
                                 px.ctypes.data_as(C.POINTER(C.c_float)), _dp(bearing), _dp(pw),
                                 ini.ctypes.data_as(capi.u8p), len(px), _dp(Tr), _dp(Tc),
                                 C.byref(prm), C.byref(nt), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle_sparse_align -> {rc}")
    return Tc.reshape(3, 4), nt.value, st.as_dict()


def align2d(img, patch_border, patch, max_iters, px):
    lib = load()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    pb = np.ascontiguousarray(patch_border, dtype=np.uint8).reshape(100)
    p = np.ascontiguousarray(patch, dtype=np.uint8).reshape(64)
    pxa = np.array(px, dtype=np.float64)
    ok = lib.oracle_align2d(img.ctypes.data_as(capi.u8p), img.shape[1], img.shape[0], img.strides[0],
                            pb.ctypes.data_as(capi.u8p), p.ctypes.data_as(capi.u8p), max_iters, _dp(pxa))
    return bool(ok), pxa


def align2d_batch(cur_pyr, patch_border, patch, level, px, max_iters):
    lib = load()
    cur, keep = capi.pyramid_struct(cur_pyr)
    pb = np.ascontiguousarray(patch_border, dtype=np.uint8)
    p = np.ascontiguousarray(patch, dtype=np.uint8)
    lv = np.ascontiguousarray(level, dtype=np.int32)
    pxa = np.array(px, dtype=np.float64).reshape(-1, 2).copy()
    m = len(lv)
    conv = np.zeros(m, dtype=np.uint8)
    rc = lib.oracle_align2d_batch(C.byref(cur), pb.ctypes.data_as(capi.u8p), p.ctypes.data_as(capi.u8p),
                                  lv.ctypes.data_as(C.POINTER(C.c_int32)), _dp(pxa),
                                  conv.ctypes.data_as(capi.u8p), max_iters, m)
    if rc != 0:
        raise RuntimeError(f"oracle_align2d_batch -> {rc}")
    return conv.astype(bool), pxa


def pyrdown(img):
    lib = load()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), dtype=np.uint8)
    lib.oracle_pyrdown(img.ctypes.data_as(capi.u8p), w, h, img.strides[0],
                       out.ctypes.data_as(capi.u8p), out.strides[0])
    return out


def warp_patches(kf_pyrs, cam, T_kf_w, T_cur_w, cand_kf, ref_px, ref_level, ref_bearing, p_world,
                 max_search_level):
    lib = load()
    n_kf = len(kf_pyrs)
    arr = (capi.Pyramid * n_kf)()
    keep = []
    for i, p in enumerate(kf_pyrs):
        s, k = capi.pyramid_struct(p)
        arr[i] = s
        keep.append(k)
    m = len(cand_kf)
    Tk = np.ascontiguousarray(T_kf_w, dtype=np.float64).reshape(n_kf, 12)
    Tc = np.ascontiguousarray(T_cur_w, dtype=np.float64).reshape(12)
    ck = np.ascontiguousarray(cand_kf, dtype=np.int32)
    rp = np.ascontiguousarray(ref_px, dtype=np.float32)
    rl = np.ascontiguousarray(ref_level, dtype=np.int32)
    rb = np.ascontiguousarray(ref_bearing, dtype=np.float64)
    pw = np.ascontiguousarray(p_world, dtype=np.float64)
    aff = np.zeros((m, 4))
    sl = np.zeros(m, dtype=np.int32)
    pb = np.zeros((m, 100), dtype=np.uint8)
    pp = np.zeros((m, 64), dtype=np.uint8)
    ip = C.POINTER(C.c_int32)
    rc = lib.oracle_warp_patches(arr, n_kf, C.byref(capi.camera_struct(cam)), _dp(Tk), _dp(Tc),
                                 ck.ctypes.data_as(ip), rp.ctypes.data_as(C.POINTER(C.c_float)),
                                 rl.ctypes.data_as(ip), _dp(rb), _dp(pw), max_search_level, m,
                                 _dp(aff), sl.ctypes.data_as(ip), pb.ctypes.data_as(capi.u8p),
                                 pp.ctypes.data_as(capi.u8p))
    if rc != 0:
        raise RuntimeError(f"oracle_warp_patches -> {rc}")
    return aff, sl, pb, pp
