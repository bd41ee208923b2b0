"""One tracked frame: the four-call chain against ONE dsdtm_track_frame (wall clock of the library calls alone, medians).
usage: python tools/track_frame_bench.py [n_points]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsdtm_amd import capi, synth, search, tracking, feature_alignment as FA
from dsdtm_amd.frame import Config, Frame
from tests.test_search_gpu import make_world

torch.cuda.init()
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 900
Config.Set("Camera.CellSize", 25); Config.Set("Camera.MaxPyraLevels", 5); Config.Set("Camera.Min_fts", 15)
ctx = capi.default_context(0)
cam, kfs, cur, mps = make_world(11, n_points=n_points)
ref = kfs[0]
nf = min(ref.n_features, 300)
bb = ref.bearing[:nf]
last = Frame(cam, ref.mvImg_Pyr, ref.Get_Pose())
last.set_features(ref.px[:nf], bb, bb * (2.0 / bb[:, 2:3]), np.ones(nf, np.uint8))
image = cur.mvImg_Pyr[0]
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
if pinned:                                  # the capture buffer of a tracker that keeps its images in pinned memory
    keep = torch.from_numpy(np.ascontiguousarray(image)).pin_memory()
    image = keep.numpy()
call = tracking.TrackCall(ctx, cam, image, 5, last, ref.Get_Pose(), (5, 0, 8, 15), 20, kfs, mps)
frame_destroy = ctx.lib.dsdtm_frame_destroy
ts = []
for k in range(80):
    t0 = time.perf_counter()
    rc = call.run_raw()
    t1 = time.perf_counter()
    assert rc == 0, ctx.lib.dsdtm_last_error(ctx.handle)
    frame_destroy(ctx.handle, C.c_void_p(call.res.frame))
    ts.append(t1 - t0)
ts = np.array(ts[10:]) * 1e3
r = call.res
print(f"dsdtm_track_frame ({'pinned image' if pinned else 'pageable image'}): median {np.median(ts):.4f} ms  min {ts.min():.4f} ms   (n_ref {nf}, points {len(mps)}, tracked {r.n_tracked}, in grid {r.n_in_grid}, "
      f"matches {r.n_matches}, full_scan {r.replay_full_scan}, po iters {r.summary.iterations})")
