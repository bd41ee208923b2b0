"""ctypes binding of the C ABI in include/dsdtm_amd.h (the drop-in boundary).

The shared library is built in-tree by `dsdtm_amd/csrc/build.py` (hipcc, gfx950) and is
the ONLY compute path: there is no CPU fallback. Loading fails loudly when the library
is missing; compute calls fail with DSDTM_ERR_NO_DEVICE when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MAX_LEVELS = 8
OK, ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_NOMEM = 0, -1, -2, -3, -4

u8p = C.POINTER(C.c_uint8)


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("f", C.c_float), ("width", C.c_int), ("height", C.c_int)]


class Pyramid(C.Structure):
    _fields_ = [("levels", C.c_int), ("data", C.c_void_p * MAX_LEVELS),
                ("width", C.c_int * MAX_LEVELS), ("height", C.c_int * MAX_LEVELS),
                ("stride", C.c_int * MAX_LEVELS)]


class AlignParams(C.Structure):
    _fields_ = [("max_level", C.c_int), ("min_level", C.c_int), ("max_iters", C.c_int),
                ("min_fts", C.c_int)]


class AlignStats(C.Structure):
    _fields_ = [("iters", C.c_int32 * MAX_LEVELS), ("n_ref", C.c_int32 * MAX_LEVELS),
                ("n_vis", C.c_int32 * MAX_LEVELS), ("exit_code", C.c_int32 * MAX_LEVELS),
                ("chi2", C.c_double * MAX_LEVELS)]

    def as_dict(self):
        return {k: list(getattr(self, k)) for k, _ in self._fields_}


STATS_DTYPE = np.dtype([("iters", "<i4", MAX_LEVELS), ("n_ref", "<i4", MAX_LEVELS),
                        ("n_vis", "<i4", MAX_LEVELS), ("exit_code", "<i4", MAX_LEVELS),
                        ("chi2", "<f8", MAX_LEVELS)])
assert STATS_DTYPE.itemsize == C.sizeof(AlignStats)


class BatchDesc(C.Structure):
    _fields_ = [("n_pairs", C.c_int), ("max_features", C.c_int), ("levels", C.c_int),
                ("width", C.c_int * MAX_LEVELS), ("height", C.c_int * MAX_LEVELS),
                ("stride", C.c_int * MAX_LEVELS), ("level_offset", C.c_size_t * MAX_LEVELS),
                ("pyr_pitch", C.c_size_t),
                ("ref_pyr", C.c_void_p), ("cur_pyr", C.c_void_p), ("px_xy", C.c_void_p),
                ("bearing", C.c_void_p), ("p_world", C.c_void_p), ("initial", C.c_void_p),
                ("n_features", C.c_void_p), ("T_ref_w", C.c_void_p), ("T_cur_w", C.c_void_p),
                ("n_tracked", C.c_void_p), ("stats", C.c_void_p)]


class StreamDesc(C.Structure):
    """dsdtm_stream_desc: a host sequence of level-0 images + feature columns for dsdtm_sparse_align_batch_streamed."""
    _fields_ = [("n_pairs", C.c_int32), ("max_features", C.c_int32), ("levels", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("row_stride", C.c_int32), ("image_pitch", C.c_size_t),
                ("ref_image", C.c_void_p), ("cur_image", C.c_void_p), ("px_xy", C.c_void_p),
                ("bearing", C.c_void_p), ("p_world", C.c_void_p), ("initial", C.c_void_p),
                ("n_features", C.c_void_p), ("T_ref_w", C.c_void_p), ("T_cur_w", C.c_void_p),
                ("n_tracked", C.c_void_p), ("stats", C.c_void_p)]


class ImageDesc(C.Structure):
    _fields_ = [("levels", C.c_int), ("width", C.c_int * MAX_LEVELS),
                ("height", C.c_int * MAX_LEVELS), ("stride", C.c_int * MAX_LEVELS),
                ("level_offset", C.c_size_t * MAX_LEVELS), ("bytes", C.c_size_t),
                ("data", C.c_void_p)]


def camera_struct(cam) -> Camera:
    return Camera(cam.fx, cam.fy, cam.cx, cam.cy, cam.f, cam.width, cam.height)


def pyramid_struct(levels) -> tuple[Pyramid, list]:
    """levels: list of 2-D uint8 arrays (rows may be strided). Returns (struct, keepalive)."""
    p = Pyramid()
    p.levels = len(levels)
    keep = []
    for i, a in enumerate(levels):
        if a.dtype != np.uint8 or a.ndim != 2 or a.strides[1] != 1:
            a = np.ascontiguousarray(a, dtype=np.uint8)
        keep.append(a)
        p.data[i] = a.ctypes.data
        p.width[i] = a.shape[1]
        p.height[i] = a.shape[0]
        p.stride[i] = a.strides[0]
    return p, keep


def pyramid_layout(width: int, height: int, levels: int, align: int = 64):
    """Packed-pyramid geometry used by the *_device entry points: per level
    (w, h, stride=w, offset); offsets aligned to `align` bytes, pitch too."""
    ws, hs, offs = [], [], []
    off = 0
    w, h = width, height
    for _ in range(levels):
        ws.append(w)
        hs.append(h)
        offs.append(off)
        off += (w * h + align - 1) // align * align
        w, h = (w + 1) // 2, (h + 1) // 2
    return ws, hs, list(ws), offs, off


def declare_signatures(lib, prefix: str, with_ctx: bool):
    """Attach argtypes for the entry points shared by the product library (prefix 'dsdtm_',
    ctx first) and the oracle (prefix 'oracle_', no ctx)."""
    ctx = [C.c_void_p] if with_ctx else []
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    f = getattr(lib, prefix + "sparse_align")
    f.restype = C.c_int
    f.argtypes = ctx + [C.POINTER(Pyramid), C.POINTER(Pyramid), C.POINTER(Camera), fp, dp, dp, u8p,
                        C.c_int, dp, dp, C.POINTER(AlignParams), C.POINTER(C.c_int),
                        C.POINTER(AlignStats)]
    f = getattr(lib, prefix + "align2d_batch")
    f.restype = C.c_int
    f.argtypes = ctx + [C.POINTER(Pyramid), u8p, u8p, ip, dp, u8p, C.c_int, C.c_int]
    f = getattr(lib, prefix + "warp_patches")
    f.restype = C.c_int
    f.argtypes = ctx + [C.POINTER(Pyramid), C.c_int, C.POINTER(Camera), dp, dp, ip, fp, ip, dp, dp,
                        C.c_int, C.c_int, dp, ip, u8p, u8p]


# every symbol include/dsdtm_amd.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "dsdtm_create", "dsdtm_destroy", "dsdtm_last_error", "dsdtm_version", "dsdtm_device_count",
    "dsdtm_sparse_align", "dsdtm_sparse_align_batch_device", "dsdtm_sparse_align_check", "dsdtm_sparse_align_workspace_bytes",
    "dsdtm_reserve", "dsdtm_align2d_batch", "dsdtm_align2d_batch_device",
    "dsdtm_pyrdown_batch_device", "dsdtm_pyrdown", "dsdtm_warp_patches",
    "dsdtm_frame_create", "dsdtm_frame_create_from_image", "dsdtm_frame_destroy", "dsdtm_sparse_align_frames",
    "dsdtm_detect_cells", "dsdtm_detect_cells_frame", "dsdtm_match_candidates_frames",
    "dsdtm_pose_optimization", "dsdtm_pose_optimization_batch_device",
    "dsdtm_sparse_align_batch_sharded", "dsdtm_sparse_align_batch_streamed", "dsdtm_shard_range", "dsdtm_detect_cells_batch_device",
    "dsdtm_match_candidates_batch_device", "dsdtm_match_candidates_scratch_bytes", "dsdtm_track_frame",
]


class DetectParams(C.Structure):
    _fields_ = [("cell_size", C.c_int32), ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("levels", C.c_int32),
                ("barrier", C.c_int32), ("detection_threshold", C.c_float)]


class PoseOptParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("reserved", C.c_int32)]


class PoseOptSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
                ("n_residual_blocks", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("x", C.c_double * 6)]

    def as_dict(self):
        return dict(iterations=self.iterations, successful_steps=self.successful_steps, termination=self.termination,
                    n_residual_blocks=self.n_residual_blocks, initial_cost=self.initial_cost,
                    final_cost=self.final_cost, x=np.array(list(self.x)))


class TrackDesc(C.Structure):
    """dsdtm_track_desc: one tracked frame (new image, Run against the last frame, the local map) for dsdtm_track_frame."""
    _fields_ = [("image", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32), ("levels", C.c_int32),
                ("ref", C.c_void_p), ("ref_px_xy", C.c_void_p), ("ref_bearing", C.c_void_p), ("ref_p_world", C.c_void_p),
                ("ref_initial", C.c_void_p), ("n_ref_features", C.c_int32), ("T_ref_w", C.c_void_p), ("T_seed", C.c_void_p),
                ("align", AlignParams), ("min_tracked", C.c_int32),
                ("kf", C.c_void_p), ("n_kf", C.c_int32), ("T_kf_w", C.c_void_p), ("n_points", C.c_int32),
                ("mp_world", C.c_void_p), ("mp_found", C.c_void_p), ("mp_bad", C.c_void_p), ("obs_offset", C.c_void_p),
                ("obs_kf", C.c_void_p), ("obs_px", C.c_void_p), ("obs_level", C.c_void_p), ("obs_bearing", C.c_void_p),
                ("mask", C.c_void_p), ("mask_stride", C.c_int32), ("cell_size", C.c_int32), ("max_pyr_levels", C.c_int32),
                ("max_matches", C.c_int32), ("align2d_iters", C.c_int32), ("pose_opt", PoseOptParams)]


class TrackMatch(C.Structure):
    _fields_ = [("cell", C.c_int32), ("point", C.c_int32), ("px", C.c_float * 2), ("level", C.c_int32)]


TRACK_MATCH_DTYPE = np.dtype([("cell", "<i4"), ("point", "<i4"), ("px", "<f4", 2), ("level", "<i4")])
assert TRACK_MATCH_DTYPE.itemsize == C.sizeof(TrackMatch)


class TrackResult(C.Structure):
    _fields_ = [("frame", C.c_void_p), ("T_run", C.c_double * 12), ("n_tracked", C.c_int32), ("lost", C.c_int32),
                ("stats", AlignStats), ("n_in_grid", C.c_int32), ("n_matches", C.c_int32), ("replay_full_scan", C.c_int32), ("reserved", C.c_int32),
                ("T_opt", C.c_double * 12),
                ("summary", PoseOptSummary)]


# dsdtm_pose_opt_termination
PO_FUNCTION_TOLERANCE, PO_PARAMETER_TOLERANCE, PO_GRADIENT_TOLERANCE, PO_MAX_ITERATIONS = 0, 1, 2, 3
PO_MIN_RADIUS, PO_INVALID_STEPS, PO_NO_RESIDUALS, PO_EVALUATION_FAILED = 4, 5, 6, 7

_LIBS = {}
LIB_NAME = "libdsdtm_amd.so"             # the release library: the product (exactly the header's symbols, no switches)
DIAG_LIB_NAME = "libdsdtm_amd_diag.so"   # the diagnostic build (build.py --diag): + dsdtm_debug_*, environment switches


def lib_path(diag: bool | None = False) -> str:
    diag = _DIAG_DEFAULT if diag is None else bool(diag)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", DIAG_LIB_NAME if diag else LIB_NAME)


class DsdtmError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"dsdtm_amd status {status}: {msg}")
        self.status = status


# Which library a Context is created from when the caller does not say: the release one. tools/ (A/B scripts that set
# DSDTM_* switches in the environment — only the diagnostic library reads them) export DSDTM_PY_DIAG=1; tests use
# `with capi.diag_default():`. This is the PYTHON binding's choice of file; the release library itself reads no environment.
_DIAG_DEFAULT = os.environ.get("DSDTM_PY_DIAG") == "1"


class diag_default:
    """`with capi.diag_default(): ...` — contexts created inside the block without an explicit `diag=` come from the
    diagnostic library (for code that builds its own Context, e.g. tools/ modules driven by a `diag` test)."""

    def __init__(self, flag: bool = True):
        self.flag = bool(flag)

    def __enter__(self):
        global _DIAG_DEFAULT
        self.old, _DIAG_DEFAULT = _DIAG_DEFAULT, self.flag
        return self

    def __exit__(self, *exc):
        global _DIAG_DEFAULT
        _DIAG_DEFAULT = self.old
        return False


def load(diag: bool | None = None):
    """Load libdsdtm_amd.so — or, diag=True, the diagnostic build beside it (tools/ and the tests marked `diag`: fault
    injection and A/B switches the release library does not have). No fallback: raises if it has not been built."""
    diag = _DIAG_DEFAULT if diag is None else bool(diag)
    if diag in _LIBS:
        return _LIBS[diag]
    path = lib_path(diag)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "dsdtm_amd has no CPU fallback.")
    lib = C.CDLL(path)
    declare_signatures(lib, "dsdtm_", with_ctx=True)
    lib.dsdtm_create.restype = C.c_int
    lib.dsdtm_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.dsdtm_destroy.restype = None
    lib.dsdtm_destroy.argtypes = [C.c_void_p]
    lib.dsdtm_last_error.restype = C.c_char_p
    lib.dsdtm_last_error.argtypes = [C.c_void_p]
    lib.dsdtm_version.restype = C.c_char_p
    lib.dsdtm_device_count.restype = C.c_int
    lib.dsdtm_sparse_align_batch_device.restype = C.c_int
    lib.dsdtm_sparse_align_batch_device.argtypes = [C.c_void_p, C.POINTER(BatchDesc), C.POINTER(Camera),
                                                    C.POINTER(AlignParams), C.c_void_p]
    lib.dsdtm_sparse_align_check.restype = C.c_int
    lib.dsdtm_sparse_align_check.argtypes = [C.c_void_p, C.c_void_p]
    lib.dsdtm_sparse_align_workspace_bytes.restype = C.c_size_t
    lib.dsdtm_sparse_align_workspace_bytes.argtypes = [C.POINTER(BatchDesc)]
    lib.dsdtm_reserve.restype = C.c_int
    lib.dsdtm_reserve.argtypes = [C.c_void_p, C.c_size_t]
    lib.dsdtm_align2d_batch_device.restype = C.c_int
    lib.dsdtm_align2d_batch_device.argtypes = [C.c_void_p, C.POINTER(ImageDesc), C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_void_p]
    lib.dsdtm_pyrdown_batch_device.restype = C.c_int
    lib.dsdtm_pyrdown_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_size_t), C.c_void_p]
    lib.dsdtm_pyrdown.restype = C.c_int
    lib.dsdtm_pyrdown.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    lib.dsdtm_frame_create.restype = C.c_int
    lib.dsdtm_frame_create.argtypes = [C.c_void_p, C.POINTER(Pyramid), C.POINTER(C.c_void_p)]
    lib.dsdtm_frame_create_from_image.restype = C.c_int
    lib.dsdtm_frame_create_from_image.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.dsdtm_frame_destroy.restype = None
    lib.dsdtm_frame_destroy.argtypes = [C.c_void_p, C.c_void_p]
    lib.dsdtm_sparse_align_frames.restype = C.c_int
    lib.dsdtm_sparse_align_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Camera), fp, dp, dp, u8p, C.c_int,
                                              dp, dp, C.POINTER(AlignParams), C.POINTER(C.c_int), C.POINTER(AlignStats)]
    ip32 = C.POINTER(C.c_int32)
    lib.dsdtm_detect_cells.restype = C.c_int
    lib.dsdtm_detect_cells.argtypes = [C.c_void_p, C.POINTER(Pyramid), u8p, C.POINTER(DetectParams), fp, ip32, ip32, ip32]
    lib.dsdtm_detect_cells_frame.restype = C.c_int
    lib.dsdtm_detect_cells_frame.argtypes = [C.c_void_p, C.c_void_p, u8p, C.POINTER(DetectParams), fp, ip32, ip32, ip32]
    lib.dsdtm_match_candidates_batch_device.restype = C.c_int
    lib.dsdtm_match_candidates_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_int,
                                                        C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                                        C.POINTER(Camera)] + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    lib.dsdtm_match_candidates_scratch_bytes.restype = C.c_size_t
    lib.dsdtm_match_candidates_scratch_bytes.argtypes = [C.c_int]
    lib.dsdtm_detect_cells_batch_device.restype = C.c_int
    lib.dsdtm_detect_cells_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int),
                                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.c_void_p,
                                                    C.POINTER(DetectParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dsdtm_match_candidates_frames.restype = C.c_int
    lib.dsdtm_match_candidates_frames.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(Camera), dp, dp,
                                                  ip32, fp, ip32, dp, dp, C.c_int, C.c_int, C.c_int, dp, ip32, u8p]
    lib.dsdtm_sparse_align_batch_sharded.restype = C.c_int
    lib.dsdtm_sparse_align_batch_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(BatchDesc), C.POINTER(Camera),
                                                     C.POINTER(AlignParams)]
    lib.dsdtm_sparse_align_batch_streamed.restype = C.c_int
    lib.dsdtm_sparse_align_batch_streamed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamDesc), C.c_int, C.POINTER(Camera),
                                                      C.POINTER(AlignParams)]
    lib.dsdtm_track_frame.restype = C.c_int
    lib.dsdtm_track_frame.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(TrackDesc), C.POINTER(TrackResult), C.c_void_p, C.c_void_p]
    lib.dsdtm_shard_range.restype = None
    lib.dsdtm_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if diag:
        lib.dsdtm_debug_set_option.restype = C.c_int
        lib.dsdtm_debug_set_option.argtypes = [C.c_char_p, C.c_int]
        lib.dsdtm_debug_get_option.restype = C.c_int
        lib.dsdtm_debug_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    _LIBS[diag] = lib
    return lib


class debug_options:
    """`with capi.debug_options(no_team=1): ...` — diagnostic switches of the DIAGNOSTIC library (kernels.h: Options) for
    the duration of a block; they act on contexts of that library only (Context(device, diag=True)). The release library
    has no switches. The diagnostic library reads the environment once, when its first context is created; tests and A/B
    tools change a switch afterwards through this."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        lib = load(diag=True)
        for k, v in self.kv.items():
            cur = C.c_int()
            if lib.dsdtm_debug_get_option(k.encode(), C.byref(cur)) != OK:
                raise KeyError(k)
            self.old[k] = cur.value
            lib.dsdtm_debug_set_option(k.encode(), int(v))
        return self

    def __exit__(self, *exc):
        lib = load(diag=True)
        for k, v in self.old.items():
            lib.dsdtm_debug_set_option(k.encode(), v)
        return False


def device_frame_of(ctx, frame):
    """The frame's pyramid on the device (uploaded on first use, cached on the Frame object)."""
    df = getattr(frame, "_device_frame", None)
    if df is None or df.ctx is not ctx or df.handle is None:
        df = frame._device_frame = DeviceFrame.from_pyramid(ctx, frame.mvImg_Pyr)
    return df


class DeviceFrame:
    """A frame's image pyramid kept on the device (dsdtm_frame): uploaded once, used by every Run the
    frame takes part in. `from_image` sends level 0 only and builds the pyramid with the library's
    bit-exact pyrDown (Frame::ComputeImagePyramid, reference src/Frame.cpp:74-81)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.handle = ctx, handle

    @classmethod
    def from_pyramid(cls, ctx: "Context", img_pyr):
        pyr, keep = pyramid_struct(img_pyr)
        h = C.c_void_p()
        ctx.check(ctx.lib.dsdtm_frame_create(ctx.handle, C.byref(pyr), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_image(cls, ctx: "Context", level0, levels: int):
        import numpy as np
        img = np.ascontiguousarray(level0, dtype=np.uint8)
        h = C.c_void_p()
        ctx.check(ctx.lib.dsdtm_frame_create_from_image(ctx.handle, img.ctypes.data_as(u8p), img.shape[1], img.shape[0],
                                                        img.strides[0], levels, C.byref(h)))
        return cls(ctx, h)

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.dsdtm_frame_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Owns one dsdtm_ctx (one per calling thread, as the reference classes are single-threaded)."""

    def __init__(self, device: int = 0, diag: bool | None = None):
        diag = _DIAG_DEFAULT if diag is None else bool(diag)
        self.lib = load(diag)
        self.diag = diag
        h = C.c_void_p()
        st = self.lib.dsdtm_create(device, C.byref(h))
        if st != OK:
            msg = self.lib.dsdtm_last_error(None)
            raise DsdtmError(st, msg.decode() if msg else "dsdtm_create failed")
        self.handle = h
        self.device = device

    def check(self, st: int):
        if st != OK:
            msg = self.lib.dsdtm_last_error(self.handle)
            raise DsdtmError(st, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.dsdtm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = {}


def default_context(device: int = 0, diag: bool | None = None) -> Context:
    diag = _DIAG_DEFAULT if diag is None else bool(diag)
    ctx = _DEFAULT_CTX.get((device, diag))
    if ctx is None:
        ctx = _DEFAULT_CTX[(device, diag)] = Context(device, diag)
    return ctx
