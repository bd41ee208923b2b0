"""One-off stress run (not part of the test-suite): many random configurations of the single-pair host
path and the batch path against the oracle, plus a long row of back-to-back launches.
Usage: python tools/soak.py [n_configs]   (MI355X)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsdtm_amd import capi, synth
from tests import helpers as H, oracle_lib

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ctx = capi.default_context(0)
bad = 0
t0 = time.time()
for seed in range(n_cfg):
    rng = np.random.default_rng(50000 + seed)
    width, height = int(rng.integers(100, 700)), int(rng.integers(90, 500))
    levels = int(rng.integers(1, 6))
    while levels > 1 and min(width, height) >> (levels - 1) < 20:
        levels -= 1
    n = int(rng.choice([16, 40, 64, 65, 128, 129, 191, 192, 193, 255, 256, 257, 300, 319, 320, 321, 447, 448, 449, 700, 704, 705, 1000, 1500]))
    max_level = int(rng.integers(1, levels + 1)); min_level = int(rng.integers(0, max_level))
    iters = int(rng.integers(1, 15))
    xi = tuple(rng.uniform(-1, 1, 6) * np.array([0.012, 0.012, 0.012, 0.006, 0.006, 0.006]) * rng.uniform(0.1, 3.0))
    sc = synth.make_scene(width=width, height=height, levels=levels, n_patches=n, seed=60000 + seed, xi=xi, margin=int(rng.integers(4, 12)),
                          T_ref_w=synth.random_pose(rng), frac_uninitial=float(rng.choice([0.0, 0.1, 0.5])), depth=float(rng.uniform(0.8, 6.0)))
    To, no, so = oracle_lib.sparse_align(sc, max_level, min_level, iters)
    Tg, ng, sg = H.gpu_sparse_align(sc, max_level, min_level, iters, ctx=ctx)
    ang, dt = synth.pose_error(Tg, To)
    ok = ang < 1e-8 and dt < 1e-8 and ng == no and sg["iters"] == so["iters"] and sg["exit_code"] == so["exit_code"] and sg["n_vis"] == so["n_vis"]
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, width, height, levels, (min_level, max_level), n, iters, ang, dt, ng, no, sg["iters"], so["iters"], flush=True)
print(f"{n_cfg} random configurations in {time.time()-t0:.1f} s: {bad} mismatches", flush=True)
