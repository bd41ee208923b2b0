"""Argument errors at the C ABI: every refused call returns DSDTM_ERR_INVALID with a message (never a crash, never a
launch), and the context keeps working afterwards. The reference has no error codes on this path (`Run`
returns 0 and logs); the status codes are the C boundary's own and are part of its contract (include/dsdtm_amd.h:40-46)."""
import ctypes as C

import numpy as np
import pytest

from dsdtm_amd import capi
from tests import helpers as H
from tests.conftest import cached_scene

pytestmark = pytest.mark.gpu


def _refused(ctx, rc, needle=None):
    assert rc == capi.ERR_INVALID, rc
    msg = ctx.lib.dsdtm_last_error(ctx.handle)
    assert msg and (needle is None or needle.encode() in msg), msg


def _clone(desc):
    return type(desc).from_buffer_copy(bytes(desc))


def test_batch_descriptor_checks(gpu_ctx, oracle):
    import torch
    from tests.test_sparse_align_gpu import _device_batch
    dev = torch.device("cuda", 0)
    W, Hh, L, N, P = 320, 240, 3, 150, 4
    scs = [cached_scene(width=W, height=Hh, levels=L, n_patches=N, seed=1700 + i, margin=12) for i in range(P)]
    t, b = _device_batch(torch, dev, scs, L, W, Hh)
    cam = capi.camera_struct(scs[0].cam)
    prm = capi.AlignParams(L, 0, 10, 15)
    lib, h = gpu_ctx.lib, gpu_ctx.handle

    def call(bd=b, cm=cam, pr=prm, ctxh=h):
        return lib.dsdtm_sparse_align_batch_device(ctxh, C.byref(bd) if bd is not None else None, C.byref(cm) if cm is not None else None,
                                                   C.byref(pr) if pr is not None else None, None)

    assert call(ctxh=None) == capi.ERR_INVALID                      # no context: nowhere to leave a message
    _refused(gpu_ctx, call(bd=None), "NULL")
    _refused(gpu_ctx, call(cm=None), "NULL")
    _refused(gpu_ctx, call(pr=None), "params")
    for mx, mn in [(L + 1, 0), (L, -1), (capi.MAX_LEVELS + 1, 0)]:
        _refused(gpu_ctx, call(pr=capi.AlignParams(mx, mn, 10, 15)), "level range")
    for field, value, needle in [("n_pairs", -1, "geometry"), ("max_features", -1, "geometry"), ("max_features", 32768, "geometry"),
                                 ("levels", 0, "level range"), ("levels", capi.MAX_LEVELS + 1, "geometry"),
                                 ("ref_pyr", None, "NULL"), ("T_cur_w", None, "NULL"), ("n_tracked", None, "NULL"), ("px_xy", None, "NULL"),
                                 ("pyr_pitch", b.pyr_pitch + 2, "aligned"), ("pyr_pitch", 0, "aligned"),
                                 ("cur_pyr", b.cur_pyr + 1, "aligned")]:
        bd = _clone(b)
        setattr(bd, field, value)
        _refused(gpu_ctx, call(bd=bd), needle)
    bd = _clone(b); bd.level_offset[L - 1] = b.pyr_pitch                 # the last level starts behind the pair's pyramid
    _refused(gpu_ctx, call(bd=bd), "does not fit")
    bd = _clone(b); bd.stride[0] = b.width[0] - 1
    _refused(gpu_ctx, call(bd=bd), "does not fit")
    # an empty batch is not an error, and after all of the above the context still computes the oracle's poses
    bd = _clone(b); bd.n_pairs = 0
    gpu_ctx.check(call(bd=bd))
    gpu_ctx.check(call())
    gpu_ctx.check(lib.dsdtm_sparse_align_check(h, None))
    Tg = t["Tc"].cpu().numpy()
    for i, sc in enumerate(scs):
        To, no, _ = oracle.sparse_align(sc, L, 0, 10)
        H.assert_pose_close(Tg[i], To, H.TIGHT_RAD * 10, H.TIGHT_M * 10, what=f"pair {i}")


def test_single_pair_and_frame_entries(gpu_ctx):
    sc = cached_scene(width=320, height=240, levels=3, n_patches=60, seed=5, margin=12)
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    ref, k1 = capi.pyramid_struct(sc.ref_pyr)
    cur, k2 = capi.pyramid_struct(sc.cur_pyr)
    cam = capi.camera_struct(sc.cam)
    prm = capi.AlignParams(3, 0, 10, 15)
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    px, bear, pw, ini = (np.ascontiguousarray(sc.px, np.float32), np.ascontiguousarray(sc.bearing), np.ascontiguousarray(sc.p_world),
                         np.ascontiguousarray(sc.initial, np.uint8))
    Tr, Tc = np.ascontiguousarray(sc.T_ref_w.reshape(12)), np.ascontiguousarray(sc.T_cur_w_seed.reshape(12))
    nt = C.c_int(0)

    def run(refp=ref, curp=cur, pxp=px, n=len(px), Tcp=Tc, pr=prm):
        return lib.dsdtm_sparse_align(h, C.byref(refp), C.byref(curp), C.byref(cam), pxp.ctypes.data_as(fp) if pxp is not None else None,
                                      bear.ctypes.data_as(dp), pw.ctypes.data_as(dp), ini.ctypes.data_as(capi.u8p), n,
                                      Tr.ctypes.data_as(dp), Tcp.ctypes.data_as(dp) if Tcp is not None else None, C.byref(pr), C.byref(nt), None)

    _refused(gpu_ctx, run(pxp=None), "NULL")
    _refused(gpu_ctx, run(Tcp=None), "NULL")
    _refused(gpu_ctx, run(pr=capi.AlignParams(4, 0, 10, 15)), "level range")
    short, k3 = capi.pyramid_struct(sc.cur_pyr[:2])
    _refused(gpu_ctx, run(curp=short), "level count")
    other, k4 = capi.pyramid_struct([l[:-2] for l in sc.cur_pyr])
    _refused(gpu_ctx, run(curp=other), "differ in size")
    bad = _clone(ref); bad.levels = 0
    _refused(gpu_ctx, run(refp=bad), "pyramid")
    gpu_ctx.check(run())
    assert nt.value > 0
    # device-resident frames belong to the context that made them
    f1, f2 = C.c_void_p(), C.c_void_p()
    gpu_ctx.check(lib.dsdtm_frame_create(h, C.byref(ref), C.byref(f1)))
    ctx2 = capi.Context(0)
    ctx2.check(lib.dsdtm_frame_create(ctx2.handle, C.byref(cur), C.byref(f2)))
    rc = lib.dsdtm_sparse_align_frames(h, f1, f2, C.byref(cam), px.ctypes.data_as(fp), bear.ctypes.data_as(dp), pw.ctypes.data_as(dp),
                                       ini.ctypes.data_as(capi.u8p), len(px), Tr.ctypes.data_as(dp), Tc.ctypes.data_as(dp), C.byref(prm),
                                       C.byref(nt), None)
    _refused(gpu_ctx, rc, "another context")
    rc = lib.dsdtm_sparse_align_frames(h, f1, None, C.byref(cam), px.ctypes.data_as(fp), bear.ctypes.data_as(dp), pw.ctypes.data_as(dp),
                                       ini.ctypes.data_as(capi.u8p), len(px), Tr.ctypes.data_as(dp), Tc.ctypes.data_as(dp), C.byref(prm),
                                       C.byref(nt), None)
    _refused(gpu_ctx, rc, "NULL")
    lib.dsdtm_frame_destroy(ctx2.handle, f2)
    lib.dsdtm_frame_destroy(h, f1)
    ctx2.close()


def test_align2d_pyramid_and_shard_entries(gpu_ctx):
    from dsdtm_amd import synth
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    tex = np.clip(np.rint(synth.make_texture(120, 160, 3)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 3)
    cur, keep = capi.pyramid_struct(pyr)
    pb, p = H.make_border_patches(pyr[0], [(60.2, 50.3), (80.0, 40.0)])
    px = np.array([[60.0, 50.0], [80.5, 40.5]])
    conv = np.zeros(2, np.uint8)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)

    def a2d(level, m=2, pbp=pb):
        lv = np.asarray(level, np.int32)
        return lib.dsdtm_align2d_batch(h, C.byref(cur), pbp.ctypes.data_as(capi.u8p) if pbp is not None else None, p.ctypes.data_as(capi.u8p),
                                       lv.ctypes.data_as(ip), px.ctypes.data_as(dp), conv.ctypes.data_as(capi.u8p), 10, m)

    _refused(gpu_ctx, a2d([0, 3]), "outside the pyramid")
    _refused(gpu_ctx, a2d([-1, 0]), "outside the pyramid")
    _refused(gpu_ctx, a2d([0, 0], m=-1), "NULL")
    _refused(gpu_ctx, a2d([0, 0], pbp=None), "NULL")
    gpu_ctx.check(a2d([0, 0], m=0))
    gpu_ctx.check(a2d([0, 0]))
    # the partition helper of the sharded entry: ceil(P / G) pairs per shard, the tail shards may be empty
    lo, hi = C.c_int(), C.c_int()
    got = []
    for g in range(3):
        lib.dsdtm_shard_range(10, 3, g, C.byref(lo), C.byref(hi))
        got.append((lo.value, hi.value))
    assert got == [(0, 4), (4, 8), (8, 10)]
    lib.dsdtm_shard_range(2, 3, 2, C.byref(lo), C.byref(hi))
    assert lo.value == hi.value


def test_detector_pyramid_pose_and_sharded_entries(gpu_ctx):
    from dsdtm_amd import synth
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    tex = np.clip(np.rint(synth.make_texture(120, 160, 9)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, 3)
    ps, keep = capi.pyramid_struct(pyr)
    G = 7 * 5
    occ = np.zeros(G, np.uint8)
    score = np.zeros(G, np.float32)
    cx, cy, cl = (np.zeros(G, np.int32) for _ in range(3))
    ip, fp, dp = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)

    def det(prm, scorep=score):
        return lib.dsdtm_detect_cells(h, C.byref(ps), occ.ctypes.data_as(capi.u8p), C.byref(prm),
                                      scorep.ctypes.data_as(fp) if scorep is not None else None, cx.ctypes.data_as(ip), cy.ctypes.data_as(ip),
                                      cl.ctypes.data_as(ip))

    good = capi.DetectParams(25, 7, 5, 3, 20, 5.0)
    for bad in [capi.DetectParams(0, 7, 5, 3, 20, 5.0), capi.DetectParams(25, 0, 5, 3, 20, 5.0), capi.DetectParams(25, 7, 5, 4, 20, 5.0),
                capi.DetectParams(25, 7, 5, 3, 255, 5.0), capi.DetectParams(25, 7, 5, 3, 20, float("nan")),
                capi.DetectParams(25, 7, 5, 3, 20, -1.0), capi.DetectParams(25, 1 << 13, 1 << 13, 3, 20, 5.0)]:
        _refused(gpu_ctx, det(bad), "detector parameters")
    _refused(gpu_ctx, det(good, scorep=None), "NULL")
    gpu_ctx.check(det(good))
    assert (score > 0).any()

    # cv::pyrDown chain: geometry that is not a pyramid, missing outputs
    lv = [np.zeros((60, 80), np.uint8), np.zeros((30, 40), np.uint8)]
    outs = (C.c_void_p * 3)(None, lv[0].ctypes.data, lv[1].ctypes.data)
    strides = (C.c_int * 3)(160, 80, 40)
    l0 = tex.ctypes.data_as(capi.u8p)
    _refused(gpu_ctx, lib.dsdtm_pyrdown(h, l0, 160, 120, 100, 3, outs, strides), "bad argument")          # stride < width
    _refused(gpu_ctx, lib.dsdtm_pyrdown(h, l0, 160, 120, 160, 9, outs, strides), "bad argument")          # more than DSDTM_MAX_LEVELS
    _refused(gpu_ctx, lib.dsdtm_pyrdown(h, None, 160, 120, 160, 3, outs, strides), "bad argument")
    _refused(gpu_ctx, lib.dsdtm_pyrdown(h, l0, 160, 120, 160, 3, None, strides), "bad argument")
    short = (C.c_int * 3)(160, 80, 39)
    _refused(gpu_ctx, lib.dsdtm_pyrdown(h, l0, 160, 120, 160, 3, outs, short), "output level 2")
    gpu_ctx.check(lib.dsdtm_pyrdown(h, l0, 160, 120, 160, 3, outs, strides))
    assert np.array_equal(lv[0], pyr[1]) and np.array_equal(lv[1], pyr[2])
    import torch
    d = torch.zeros(40000, dtype=torch.uint8, device="cuda:0")
    w3, h3, s3 = (C.c_int * 3)(160, 80, 41), (C.c_int * 3)(120, 60, 30), (C.c_int * 3)(160, 80, 41)
    off = (C.c_size_t * 3)(0, 19200, 24000)
    _refused(gpu_ctx, lib.dsdtm_pyrdown_batch_device(h, d.data_ptr(), 40000, 1, 3, w3, h3, s3, off, None), "is not ((w+1)/2")
    w3[2] = s3[2] = 40
    _refused(gpu_ctx, lib.dsdtm_pyrdown_batch_device(h, d.data_ptr(), 25000, 1, 3, w3, h3, s3, off, None), "does not fit")
    _refused(gpu_ctx, lib.dsdtm_pyrdown_batch_device(h, None, 40000, 1, 3, w3, h3, s3, off, None), "bad argument")
    gpu_ctx.check(lib.dsdtm_pyrdown_batch_device(h, d.data_ptr(), 40000, 1, 3, w3, h3, s3, off, None))
    torch.cuda.synchronize()

    # pose refinement: a used feature on a level the pyramid cannot have; NULL outputs
    n = 30
    bear, pw = np.zeros((n, 3)), np.ones((n, 3))
    bear[:, 2] = 1.0
    level, use = np.zeros(n, np.int32), np.ones(n, np.uint8)
    T = np.ascontiguousarray(np.eye(4)[:3].reshape(12))
    norms = np.zeros(n)
    summ = capi.PoseOptSummary()
    prm = capi.PoseOptParams(5, 0)

    lib.dsdtm_pose_optimization.restype = C.c_int
    lib.dsdtm_pose_optimization.argtypes = [C.c_void_p, dp, dp, ip, capi.u8p, C.c_int, dp, C.POINTER(capi.PoseOptParams), dp,
                                            C.POINTER(capi.PoseOptSummary)]

    def po(lvl=level, Tp=T, sm=summ):
        return lib.dsdtm_pose_optimization(h, bear.ctypes.data_as(dp), pw.ctypes.data_as(dp), lvl.ctypes.data_as(ip), use.ctypes.data_as(capi.u8p), n,
                                           Tp.ctypes.data_as(dp) if Tp is not None else None, C.byref(prm), norms.ctypes.data_as(dp),
                                           C.byref(sm) if sm is not None else None)

    lv_bad = level.copy(); lv_bad[7] = capi.MAX_LEVELS
    _refused(gpu_ctx, po(lvl=lv_bad), "feature 7")
    _refused(gpu_ctx, po(Tp=None), "NULL")
    _refused(gpu_ctx, po(sm=None), "NULL")

    # the sharded entry wants one context per shard and host pointers
    sc = cached_scene(width=320, height=240, levels=3, n_patches=60, seed=5, margin=12)
    hb = capi.BatchDesc()                                                # all pointers NULL
    hb.n_pairs, hb.max_features, hb.levels = 1, 60, 3
    cam = capi.camera_struct(sc.cam)
    ap = capi.AlignParams(3, 0, 10, 15)
    two = (C.c_void_p * 2)(h, h)
    _refused(gpu_ctx, lib.dsdtm_sparse_align_batch_sharded(two, 1, C.byref(hb), C.byref(cam), C.byref(ap)), "NULL host pointers")
    assert lib.dsdtm_sparse_align_batch_sharded(two, 0, C.byref(hb), C.byref(cam), C.byref(ap)) == capi.ERR_INVALID
    assert lib.dsdtm_sparse_align_batch_sharded(None, 2, C.byref(hb), C.byref(cam), C.byref(ap)) == capi.ERR_INVALID
