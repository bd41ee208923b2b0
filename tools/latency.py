"""Single-pair latency of the HOST entry points (PCIe-inclusive): what Tracking would see per
frame through the reference-shaped classes. Configs 2, 3 and 5 of BASELINE.md.
Usage: python tools/latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsdtm_amd import capi, synth, feature_alignment as FA
from dsdtm_amd.frame import Config, frames_from_scene
from dsdtm_amd.sparse_align import Sprase_ImgAlign
from tests import helpers, oracle_lib

ctx = capi.default_context(0)
Config.Set("Camera.Min_fts", 15)
for name, kw, prm in [("config2 640x480 N=300", dict(), (4, 0, 10)),
                      ("config3 640x480 N=1000", dict(n_patches=1000, cam=synth.Camera.tum(640, 480, synth.TUM_FR3)), (4, 0, 10)),
                      ("config5 1280x960 N=2000", dict(width=1280, height=960, n_patches=2000, margin=60), (4, 0, 10))]:
    sc = synth.make_scene(**kw)
    al = Sprase_ImgAlign(*prm, ctx=ctx)
    ts = []
    for i in range(30):
        cur, ref = frames_from_scene(sc)
        t0 = time.perf_counter(); n = al.Run(cur, ref); ts.append(time.perf_counter() - t0)
    # the same Run with both frames resident on the device (dsdtm_frame): features and poses only cross PCIe
    alr = Sprase_ImgAlign(*prm, ctx=ctx, resident_frames=True)
    cur_r, ref_r = frames_from_scene(sc)
    tr, tu = [], []
    for i in range(30):
        seed = frames_from_scene(sc)[0].Get_Pose()
        cur_r.Set_Pose(seed)
        t0 = time.perf_counter(); nr = alr.Run(cur_r, ref_r); tr.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); df = capi.DeviceFrame.from_image(ctx, sc.cur_pyr[0], len(sc.cur_pyr)); tu.append(time.perf_counter() - t0)
        df.close()
    assert nr == n and np.array_equal(cur_r.Get_Pose(), cur.Get_Pose())
    t0 = time.perf_counter(); To, no, so = oracle_lib.sparse_align(sc, *prm); tc = time.perf_counter() - t0
    ang, dt = synth.pose_error(cur.Get_Pose(), To)
    print(f"{name}: GPU host call median {np.median(ts[5:])*1e3:.3f} ms (min {min(ts)*1e3:.3f}) | resident frames "
          f"{np.median(tr[5:])*1e3:.3f} ms per Run + {np.median(tu[5:])*1e3:.3f} ms per new frame (level 0 upload + device pyrDown) | "
          f"CPU oracle {tc*1e3:.2f} ms | "
          f"n={n}/{no} iters={al.last_stats['iters'][:4]} delta={ang:.1e} rad {dt:.1e} m")
# config 5 second half: 2000 Align2D problems on the 1280x960 current frame
sc = synth.make_scene(width=1280, height=960, n_patches=2000, margin=60)
rng = np.random.default_rng(0)
img = sc.cur_pyr[0]
cs = np.stack([rng.uniform(20, 1260, 2000), rng.uniform(20, 940, 2000)], 1)
pb, p = helpers.make_border_patches(img, cs)
px0 = cs + rng.uniform(-1.5, 1.5, cs.shape)
ts = []
for i in range(20):
    t0 = time.perf_counter(); cg, pxg = FA.align2d_batch(sc.cur_pyr, pb, p, np.zeros(2000, np.int32), px0, 10, ctx=ctx); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); co, pxo = oracle_lib.align2d_batch(sc.cur_pyr, pb, p, np.zeros(2000, np.int32), px0, 10); tc = time.perf_counter() - t0
same = cg == co
print(f"config5 Align2D x2000: GPU host call median {np.median(ts[3:])*1e3:.3f} ms | CPU oracle {tc*1e3:.2f} ms | flags equal {same.mean():.4f} "
      f"max|dpx| {np.abs(pxg-pxo)[same & co].max():.2e} converged {co.mean():.3f}")
# keyframe creation: Feature_detector::detect on a 640x480, 5-level frame (src/Tracking.cpp:416)
from dsdtm_amd.feature_detection import Feature_detector
from dsdtm_amd.frame import Frame
img = np.clip(np.rint(synth.make_texture(480, 640, 99)), 0, 255).astype(np.uint8)
pyr5 = synth.build_pyramid(img, 5)
det = Feature_detector(640, 480, ctx=ctx)
fr = Frame(synth.Camera.tum(640, 480), pyr5)
th, tf = [], []
for i in range(20):
    t0 = time.perf_counter(); cells = det.detect_cells(fr, 5.0); th.append(time.perf_counter() - t0)
fr._device_frame = capi.DeviceFrame.from_pyramid(ctx, pyr5)
for i in range(20):
    t0 = time.perf_counter(); cells_f = det.detect_cells(fr, 5.0); tf.append(time.perf_counter() - t0)
t0 = time.perf_counter(); want = oracle_lib.detect_cells(pyr5, 5, det.mCell_size, det.mGrid_cols, det.mGrid_rows, None, 5.0); tc = time.perf_counter() - t0
t0 = time.perf_counter(); n_new = det.detect(fr, 5.0); td = time.perf_counter() - t0
same = all(np.array_equal(a, b) for a, b in zip(cells, want)) and all(np.array_equal(a, b) for a, b in zip(cells_f, want))
print(f"detector 640x480 x5 levels: per-cell corners GPU host call median {np.median(th[3:])*1e3:.3f} ms | resident frame "
      f"{np.median(tf[3:])*1e3:.3f} ms | CPU oracle {tc*1e3:.1f} ms | equal {same} | full detect() incl. host bookkeeping {td*1e3:.2f} ms, {n_new} features")
