"""One-off stress run (not part of the test-suite) of the team kernel's inter-workgroup exchange: thousands of
single-pair Runs with feature counts that alternate between team sizes, every result compared bit for bit with
the first one of its size. Usage: python tools/soak_team.py [rounds]   (MI355X)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsdtm_amd import capi, synth
from dsdtm_amd.frame import Config, frames_from_scene
from dsdtm_amd.sparse_align import Sprase_ImgAlign

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
ctx = capi.default_context(0)
Config.Set("Camera.Min_fts", 15)
cases = []
for n in (705, 1000, 1537, 2000, 3000, 4096, 5000, 9000):
    sc = synth.make_scene(n_patches=n, seed=5)
    al = Sprase_ImgAlign(4, 0, 10, ctx=ctx, resident_frames=True)
    cur, ref = frames_from_scene(sc)
    cases.append((n, sc, al, cur, ref, None))
bad = 0
t0 = time.time()
for r in range(rounds):
    for i, (n, sc, al, cur, ref, first) in enumerate(cases):
        cur.Set_Pose(sc.T_cur_w_seed)
        nt = al.Run(cur, ref)
        res = (nt, cur.Get_Pose().tobytes())
        if first is None:
            cases[i] = (n, sc, al, cur, ref, res)
        elif res != first:
            bad += 1
            print("MISMATCH round", r, "N", n, flush=True)
print(f"{rounds} rounds x {len(cases)} team sizes in {time.time()-t0:.1f} s: {bad} results differ from the first of their size", flush=True)
