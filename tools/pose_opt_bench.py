"""Pose-only refinement (Optimizer::PoseOptimization) on the MI355X: device-resident batch
throughput (HIP events on the launch stream), one-frame latency through the host entry point
(PCIe-inclusive), and the CPU restatement on this host beside it.
Usage: python tools/pose_opt_bench.py [frames] [features] [nolatency]   (MI355X)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsdtm_amd import capi, synth
from dsdtm_amd.optimizer import pose_optimization
from tests import oracle_lib as O

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0); ctx = capi.Context(0); st = torch.cuda.Stream(device=dev)
base = 64                                   # distinct synthetic frames, tiled to F
probs = [synth.make_pose_problem(1000 + k, n=N, max_level=3) for k in range(base)]
rep = (F + base - 1) // base
stack = lambda f: np.concatenate([np.stack([f(p) for p in probs])] * rep)[:F]
bearing, pw, level, use = stack(lambda p: p.bearing), stack(lambda p: p.p_world), stack(lambda p: p.level), stack(lambda p: p.use)
T0 = stack(lambda p: p.T_seed.reshape(12))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_b, d_p, d_l, d_u, d_T0 = t(bearing), t(pw), t(level), t(use), t(T0)
d_T = d_T0.clone()
d_rn = torch.zeros((F, N), dtype=torch.float64, device=dev)
d_sm = torch.zeros((F, C.sizeof(capi.PoseOptSummary)), dtype=torch.uint8, device=dev)
prm = capi.PoseOptParams(100, 0)
f = ctx.lib.dsdtm_pose_optimization_batch_device
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(capi.PoseOptParams), C.c_void_p, C.c_void_p, C.c_void_p]

def launch():
    with torch.cuda.stream(st):
        d_T.copy_(d_T0, non_blocking=True)
    ctx.check(f(ctx.handle, F, N, None, d_b.data_ptr(), d_p.data_ptr(), d_l.data_ptr(), d_u.data_ptr(), d_T.data_ptr(), C.byref(prm),
                d_rn.data_ptr(), d_sm.data_ptr(), st.cuda_stream))

for _ in range(5): launch()
st.synchronize()
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps): launch()
e1.record(st); st.synchronize()
ms = e0.elapsed_time(e1) / reps
sms = [capi.PoseOptSummary.from_buffer_copy(r.tobytes()) for r in d_sm.cpu().numpy()]
its = np.array([s.iterations for s in sms]); blocks = np.array([s.n_residual_blocks for s in sms])
evals = its + 1                                 # every iteration evaluates its candidate once, plus the initial one
print(f"pose refinement, batch: {F} frames x {N} features ({blocks.mean():.0f} residual blocks), iterations mean {its.mean():.1f} "
      f"(min {its.min()}, max {its.max()})")
print(f"  {ms:.3f} ms per launch -> {F/ms*1e3/1e3:.1f} k refinements/s, {(evals*blocks).sum()/ms*1e3/1e9:.2f} G block evaluations/s")
flop_eval = 330                                 # FP64 flops per block evaluation, divisions counted as one
print(f"  ~{(evals*blocks).sum()*flop_eval/ms*1e3/1e12:.2f} TFLOP/s FP64 of ~78 (vector peak): latency-bound, one wave per frame")

if len(sys.argv) > 3 and sys.argv[3] == "nolatency":      # profiling runs: the batch launches only
    sys.exit(0)
# one frame through the host entry point (what Tracking::TrackWithLocalMap would call)
P = probs[0]
T = np.ascontiguousarray(P.T_seed).reshape(12).copy()
for _ in range(20): T[:] = P.T_seed.reshape(12); rn, sm = pose_optimization(ctx, P.bearing, P.p_world, P.level, P.use, T)
ts = []
for _ in range(200):
    T[:] = P.T_seed.reshape(12)
    t0 = time.perf_counter(); rn, sm = pose_optimization(ctx, P.bearing, P.p_world, P.level, P.use, T); ts.append(time.perf_counter() - t0)
print(f"one frame through dsdtm_pose_optimization (H2D + kernel + D2H, {sm['iterations']} iterations): median {np.median(ts)*1e3:.3f} ms")
# the CPU restatement on this host, one thread
for ls, name in ((0, "Householder QR (Ceres DENSE_QR form)"), (1, "normal equations")):
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 2.0:
        Pk = probs[k % base]; O.pose_optimization(Pk.bearing, Pk.p_world, Pk.level, Pk.use, Pk.T_seed, linear_solver=ls); k += 1
    dt = (time.perf_counter() - t0) / k
    print(f"CPU restatement, {name}: {dt*1e3:.3f} ms per frame, 1 thread -> {1/dt:.0f} refinements/s")
