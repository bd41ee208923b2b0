"""One-off stress run (not part of the test-suite) of the two-member workspace kernel (1025..2048 patches in batches:
one pair on two compute units, partials exchanged through tagged words): many launches of a batch larger than the GPU
holds at once, several launches in flight on different streams, every result compared bit for bit with the first
launch's. A wait that runs out would raise the timeout flag (dsdtm_sparse_align_check).
Usage: python tools/soak_duo.py [rounds]   (MI355X)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsdtm_amd import capi, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0); ctx = capi.Context(0)
cam = synth.Camera.tum(640, 480); cs = capi.camera_struct(cam); prm = capi.AlignParams(4, 0, 10, 15)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
d = bench.build_batch(torch, dev, ctx, cam, 700, 640, 480, 4, 2000, seed=77, stream=streams[0])      # 1400 workgroups on 256 CUs
outs = [dict(T=d["T_seed"].clone(), nt=torch.zeros_like(d["n_tracked"])) for _ in streams]
descs = []
for o in outs:
    b = capi.BatchDesc.from_buffer_copy(bytes(d["desc"]))
    b.T_cur_w, b.n_tracked, b.stats = o["T"].data_ptr(), o["nt"].data_ptr(), None
    descs.append(b)
first = None
t0 = time.time()
for r in range(rounds):
    for o in outs:
        o["T"].copy_(d["T_seed"])
    torch.cuda.synchronize()
    for b, st in zip(descs, streams):
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(b), C.byref(cs), C.byref(prm), st.cuda_stream))
    for st in streams:
        ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, st.cuda_stream))
    for o in outs:
        T, nt = o["T"].cpu().numpy(), o["nt"].cpu().numpy()
        if first is None:
            first = (T.copy(), nt.copy())
        assert np.array_equal(T, first[0]) and np.array_equal(nt, first[1]), f"round {r}: results differ"
err = np.array([synth.pose_error(first[0][i], d["T_true"][i]) for i in range(len(first[0]))])
print(f"{rounds} rounds x {len(streams)} launches in flight of 700 pairs x 2000 patches (two members per pair): bit-identical results, "
      f"no wait ran out, {time.time() - t0:.1f} s; median error vs ground truth {np.median(err[:, 0]):.2e} rad / {np.median(err[:, 1]):.2e} m")
