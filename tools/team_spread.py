"""Team placement by member count: members of a team on ONE XCD (workgroup stride a multiple of 8) against members
spread over all XCDs (odd stride), `Run` on resident frames. Decides Options::team_spread_min (sparse_align.hip:
team_pairs_pad). Usage: python tools/team_spread.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dsdtm_amd import capi, synth
from dsdtm_amd.frame import Config, frames_from_scene
from dsdtm_amd.sparse_align import Sprase_ImgAlign
ctx = capi.default_context(0, diag=True)   # the diagnostic library (switches)
Config.Set("Camera.Min_fts", 15)
for n in (1000, 2000, 3000, 4096, 5120, 6144, 7168, 8192):
    sc = synth.make_scene(n_patches=n, seed=5)
    alr = Sprase_ImgAlign(4, 0, 10, ctx=ctx, resident_frames=True)
    cur_r, ref_r = frames_from_scene(sc)
    row = []
    for spread_min in (65, 2):           # 65: always one XCD (teams of up to 32); 2: always spread
        with capi.debug_options(team_spread_min=spread_min):
            ts = []
            for i in range(60):
                cur_r.Set_Pose(sc.T_cur_w_seed)
                t0 = time.perf_counter(); nt = alr.Run(cur_r, ref_r); ts.append(time.perf_counter() - t0)
            row.append(np.median(ts[15:]) * 1e3)
    print(f"N={n} ({(n + 255) // 256} members): one XCD {row[0]:.4f} ms, spread {row[1]:.4f} ms", flush=True)
