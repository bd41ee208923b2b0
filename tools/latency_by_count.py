import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/dsdtm_amd") else os.getcwd())
from dsdtm_amd import capi, synth
from dsdtm_amd.frame import Config, frames_from_scene
from dsdtm_amd.sparse_align import Sprase_ImgAlign
ctx = capi.default_context(0)
Config.Set("Camera.Min_fts", 15)
for n in (300, 448, 500, 600, 704, 705, 1000, 1400, 2000, 2400, 3000, 4096, 4097, 6000, 8192, 8193, 12000, 16384, 16385):
    sc = synth.make_scene(n_patches=n, seed=5)
    alr = Sprase_ImgAlign(4, 0, 10, ctx=ctx, resident_frames=True)
    cur_r, ref_r = frames_from_scene(sc)
    ts = []
    for i in range(40):
        cur_r.Set_Pose(sc.T_cur_w_seed)
        t0 = time.perf_counter(); nt = alr.Run(cur_r, ref_r); ts.append(time.perf_counter() - t0)
    print(f"N={n}: Run on resident frames median {np.median(ts[10:])*1e3:.3f} ms (tracked {nt})", flush=True)
