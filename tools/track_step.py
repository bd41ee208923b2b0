"""One tracking step of DSDTM's front end on device-resident frames (src/Tracking.cpp:45-145): what the GPU
side costs per frame when the library replaces Frame::ComputeImagePyramid, Sprase_ImgAlign::Run,
FindMatchDirect (all candidates of SearchLocalPoints), Optimizer::PoseOptimization and Feature_detector::detect. Library-call times
(host wall clock around the C ABI calls, medians); the reference-shaped bookkeeping around them is
Python here and not timed. CPU column: the oracle's restatement of the same steps, one thread.
Usage: python tools/track_step.py   (MI355X)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsdtm_amd import capi, synth, search, feature_alignment as FA
from dsdtm_amd.frame import Config, Frame
from dsdtm_amd.sparse_align import Sprase_ImgAlign
from dsdtm_amd.feature_detection import Feature_detector
from tests import oracle_lib
from tests.test_search_gpu import make_world

def med(f, n=15, warm=3):
    ts = []
    for i in range(n + warm):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return np.median(ts[warm:]) * 1e3, r

ctx = capi.default_context(0)
Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
cam, kfs, cur, mps = make_world(11, n_points=900)
L = 5
# reference frame of Run = keyframe 0 with 3-D points behind its features (plane at 2 m, world = its camera frame)
ref = kfs[0]
nf = min(ref.n_features, 300)
b = ref.bearing[:nf]
ref_run = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
ref_run.set_features(ref.px[:nf], b, b * (2.0 / b[:, 2:3]), np.ones(nf, np.uint8))
rows = []
# 1. new frame -> device, pyramid on the device
t, df = med(lambda: capi.DeviceFrame.from_image(ctx, cur.mvImg_Pyr[0], L))
t_cpu, _ = med(lambda: [oracle_lib.pyrdown(cur.mvImg_Pyr[l]) for l in range(L - 1)], n=5, warm=1)
rows.append(("new frame: level 0 upload + 4x pyrDown", t, t_cpu))
cur._device_frame = df
capi.device_frame_of(ctx, ref_run)
for k in kfs: capi.device_frame_of(ctx, k)
# 2. Sprase_ImgAlign::Run(cur, last)
al = Sprase_ImgAlign(L, 0, 8, ctx=ctx, resident_frames=True)
seed = ref.Get_Pose().copy()
def run():
    cur.Set_Pose(seed); return al.Run(cur, ref_run)
t, n_tr = med(run)
class _S: pass
sc = _S(); sc.cam = cam; sc.ref_pyr = ref_run.mvImg_Pyr; sc.cur_pyr = cur.mvImg_Pyr; sc.px = ref_run.px; sc.bearing = ref_run.bearing
sc.p_world = ref_run.p_world; sc.initial = ref_run.initial; sc.T_ref_w = ref_run.Get_Pose(); sc.T_cur_w_seed = seed
t_cpu, _ = med(lambda: oracle_lib.sparse_align(sc, L, 0, 8), n=5, warm=1)
rows.append((f"Sprase_ImgAlign::Run ({nf} features, 5 levels, cap 8; tracked {n_tr})", t, t_cpu))
# 3. FindMatchDirect for every candidate of SearchLocalPoints
s = search.LocalPointSearch(cam, ctx=ctx, resident_frames=True)
s.ResetGrid()
for mp in mps: s.ReprojectPoint(cur, mp)
cand = []
for ci, cell in enumerate(s.mCells):
    for pos, (mp, px) in enumerate(cell):
        if mp.IsBad(): continue
        obs = search.get_closest_obs(mp, cur, kfs)
        if obs is None: continue
        cand.append((mp, px, obs[0], obs[1]))
ck = np.array([c[2] for c in cand], np.int32)
rp = np.array([kfs[c[2]].px[c[3]] for c in cand], np.float32); rl = np.array([kfs[c[2]].level[c[3]] for c in cand], np.int32)
rb = np.array([kfs[c[2]].bearing[c[3]] for c in cand]); pw = np.array([c[0].Get_Pose() for c in cand]); cpx = np.array([c[1] for c in cand])
Tk = np.array([k.Get_Pose() for k in kfs])
t, (conv, pxo, sl) = med(lambda: FA.match_candidates_frames(cur, kfs, cam, Tk, cur.Get_Pose(), ck, rp, rl, rb, pw, cpx, L - 3, 10, ctx=ctx))
def cpu_match():
    aff, sl_o, pb, pp = oracle_lib.warp_patches([k.mvImg_Pyr for k in kfs], cam, Tk, cur.Get_Pose(), ck, rp, rl, rb, pw, L - 3)
    return oracle_lib.align2d_batch(cur.mvImg_Pyr, pb, pp, sl_o, cpx / (1 << sl_o)[:, None], 10)
t_cpu, (conv_o, _) = med(cpu_match, n=3, warm=1)
rows.append((f"FindMatchDirect x {len(cand)} candidates (warp prelude + Align2D; {int(conv.sum())} converged, CPU {int(conv_o.sum())})", t, t_cpu))
# 4. Optimizer::PoseOptimization on the matched map points (src/Tracking.cpp:236)
from dsdtm_amd.optimizer import pose_optimization
ok = conv.astype(bool)
bear = synth.bearing_from_px(cam, pxo[ok].astype(np.float32)); lvl = sl[ok].astype(np.int32); pws = pw[ok]; use = np.ones(int(ok.sum()), np.uint8)
def po():
    T = np.ascontiguousarray(cur.Get_Pose(), np.float64).reshape(12).copy()
    return pose_optimization(ctx, bear, pws, lvl, use, T)
t, (rn, sm) = med(po)
t_cpu, _ = med(lambda: oracle_lib.pose_optimization(bear, pws, lvl, use, cur.Get_Pose(), linear_solver=0), n=5, warm=1)
rows.append((f"Optimizer::PoseOptimization ({int(ok.sum())} map-point observations, {sm['iterations']} trust-region iterations)", t, t_cpu))
# 5. keyframe: Feature_detector::detect (per-cell corners)
det = Feature_detector(cam.width, cam.height, ctx=ctx)
t, cells = med(lambda: det.detect_cells(cur, 5.0))
t_cpu, _ = med(lambda: oracle_lib.detect_cells(cur.mvImg_Pyr, L, det.mCell_size, det.mGrid_cols, det.mGrid_rows, None, 5.0), n=3, warm=1)
rows.append(("Feature_detector::detect, per-cell corners (keyframes only)", t, t_cpu))
print("step | GPU library call, ms | CPU oracle (1 thread), ms")
for name, g, c in rows: print(f"{name} | {g:.3f} | {c:.2f}")
print(f"per tracked frame (steps 1-4): {sum(r[1] for r in rows[:4]):.3f} ms on the GPU vs {sum(r[2] for r in rows[:4]):.1f} ms for the CPU restatement")
