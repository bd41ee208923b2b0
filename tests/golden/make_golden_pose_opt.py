"""Generates tests/golden/pose_opt.npz: inputs and expected outputs of Optimizer::PoseOptimization
(src/Optimizer.cpp:20-101) on synthetic frames.

The reference holds no expected outputs for this function and cannot be built here (Ceres, Sophus,
Eigen and OpenCV are absent), so these vectors come from the CPU restatement
oracle/pose_opt_oracle.c (Householder-QR form) after it was held against the independent numpy
restatement tests/pose_opt_restatement.py — parity unpinned, see DESIGN.md §5. They pin the
restatement against regressions and travel to the GPU box as data.

    python tests/golden/make_golden_pose_opt.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dsdtm_amd import synth            # noqa: E402
from tests import oracle_lib as O      # noqa: E402

CASES = [dict(seed=11, n=200, max_level=3), dict(seed=12, n=200, max_level=0), dict(seed=13, n=700, max_level=4, outlier_frac=0.15),
         dict(seed=14, n=40, max_level=2, noise_px=1.0), dict(seed=15, n=300, max_level=3, seed_t=0.1, seed_w=0.08),
         dict(seed=16, n=64, max_level=1, unused_frac=0.5)]

out = {"n_cases": len(CASES)}
for k, kw in enumerate(CASES):
    P = synth.make_pose_problem(**kw)
    T, rn, sm = O.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=0)
    out.update({f"bearing{k}": P.bearing, f"p_world{k}": P.p_world, f"level{k}": P.level, f"use{k}": P.use,
                f"T_seed{k}": P.T_seed, f"T_out{k}": T, f"residual_norm{k}": rn,
                f"summary{k}": np.array([sm["iterations"], sm["successful_steps"], sm["termination"], sm["n_residual_blocks"]]),
                f"cost{k}": np.array([sm["initial_cost"], sm["final_cost"]]), f"x{k}": sm["x"]})
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_opt.npz"), **out)
print("wrote pose_opt.npz:", [(int(out[f"summary{k}"][0]), int(out[f"summary{k}"][2])) for k in range(len(CASES))])
