import sys; sys.path.insert(0,'.')
import numpy as np
from dsdtm_amd import capi, synth
from tests import helpers as H, oracle_lib as O
ctx = capi.default_context(0)
for kw in [dict(width=320, height=240, levels=3, n_patches=140, seed=77, margin=12, frac_uninitial=0.05),
           dict(width=320, height=240, levels=3, n_patches=120, seed=1234, margin=12),
           dict(width=320, height=240, levels=3, n_patches=300, seed=5, margin=12)]:
    sc = synth.make_scene(**kw)
    To, no, so = O.sparse_align(sc, 3, 0, 10)
    for rep in range(3):
        Tg, ng, sg = H.gpu_sparse_align(sc, 3, 0, 10, ctx=ctx)
        print(kw['n_patches'], rep, "iters", sg['iters'][:3], so['iters'][:3], "exit", sg['exit_code'][:3], so['exit_code'][:3], "nvis", sg['n_vis'][:3], so['n_vis'][:3], "chi2", np.round(sg['chi2'][:3],6), np.round(so['chi2'][:3],6), "delta", synth.pose_error(Tg, To))
