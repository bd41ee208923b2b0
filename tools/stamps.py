"""Diagnostic: where a Gauss-Newton iteration spends its cycles (in-kernel stamps, separate
kernel instantiation; never used for timing claims). Usage: python tools/stamps.py [--pairs N]"""
import argparse, ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsdtm_amd import capi, synth

ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=1024); a = ap.parse_args()
dev = torch.device("cuda", 0); ctx = capi.Context(0, diag=True)   # the diagnostic library (dsdtm_debug_* / switches)
cam = synth.Camera.tum(640, 480); cs = capi.camera_struct(cam); prm = capi.AlignParams(4, 0, 10, 15)
st = torch.cuda.Stream(device=dev)
d = bench.build_batch(torch, dev, ctx, cam, a.pairs, 640, 480, 4, 300, seed=0xD5D7, stream=st)
stamps = torch.zeros((a.pairs * 52,), dtype=torch.int64, device=dev)
f = ctx.lib.dsdtm_debug_sparse_align_stamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(capi.BatchDesc), C.POINTER(capi.Camera), C.POINTER(capi.AlignParams), C.c_void_p, C.c_void_p]
for rep in range(3):
    d["T_cur_w"].copy_(d["T_seed"]); stamps.zero_(); torch.cuda.synchronize()
    ctx.check(f(ctx.handle, C.byref(d["desc"]), C.byref(cs), C.byref(prm), stamps.data_ptr(), st.cuda_stream)); st.synchronize()
allst = stamps.cpu().numpy().astype(np.float64)
s = allst[:a.pairs*8].reshape(a.pairs, 8)
w = allst[a.pairs*8:a.pairs*20].reshape(a.pairs, 12)
pw = allst[a.pairs*20:a.pairs*36].reshape(a.pairs, 16)
lv = allst[a.pairs*36:a.pairs*44].reshape(a.pairs, 8)
bf = allst[a.pairs*44:a.pairs*48].reshape(a.pairs, 4)
rf = allst[a.pairs*48:a.pairs*49]
rt = allst[a.pairs*49:a.pairs*52].reshape(a.pairs, 3)
n_it = s[:, 3]
print("pairs", a.pairs, "iterations/pair mean", n_it.mean())
print("cycles per block: total %.0f | first-pass waits (4 levels) %.0f | later-pass waits %.0f | solve %.0f" % (s[:,4].mean(), s[:,0].mean(), s[:,1].mean(), s[:,2].mean()))
print("per first pass %.0f | per later pass %.0f | per solve %.0f cycles" % ((s[:,0]/4).mean(), (s[:,1]/np.maximum(n_it-4,1)).mean(), (s[:,2]/n_it).mean()))
print("solve sub-phases per solve: sums+LDS %.0f | factor/apply %.0f | exp+compose+publish %.0f" % ((s[:,5]/n_it).mean(), (s[:,6]/n_it).mean(), (s[:,7]/n_it).mean()))
print("wave0: precompute %.0f (per level %.0f) | passes %.0f (per pass %.0f) | H-block+store %.0f | barriers(wait for solver) %.0f" % (w[:,0].mean(), (w[:,0]/4).mean(), w[:,1].mean(), (w[:,1]/n_it).mean(), w[:,2].mean(), w[:,3].mean()))
print("wave0 pass cycles by level 0..3: " + " | ".join("%.0f (%.1f passes)" % ((w[:,4+l]/np.maximum(w[:,8+l],1)).mean(), w[:,8+l].mean()) for l in range(4)))
print("per patch wave, cycles per iteration: pass " + " ".join("%.0f" % (pw[:,k]/n_it).mean() for k in range(5)) + " | wait for solver " + " ".join("%.0f" % (pw[:,8+k]/n_it).mean() for k in range(5)))
print("wave0 per level 0..3: precompute " + " ".join("%.0f" % lv[:,l].mean() for l in range(4)) + " | first pass " + " ".join("%.0f" % lv[:,4+l].mean() for l in range(4)))
print("wave0 wait for the solver after the FIRST pass of levels 0..3: " + " ".join("%.0f" % bf[:,l].mean() for l in range(4)) + " (later passes: %.0f)" % ((w[:,3] - bf.sum(1)) / np.maximum(n_it - 4, 1)).mean())
print("solver: H sum + factorisation + H^+ per level %.0f cycles" % (rf / 4).mean())
# pair-duration spread and what it costs a 2-pairs-per-slot launch (list scheduling on 512 slots)
dur = s[:, 4]
print("pair cycles: mean %.0f | p5 %.0f | p50 %.0f | p95 %.0f | max %.0f" % (dur.mean(), *np.percentile(dur, [5, 50, 95]), dur.max()))
def makespan(order, slots=512):
    import heapq
    h = [0.0] * slots
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + dur[i])
    return max(h)
n = len(dur)
print("list-scheduling makespan / (2*mean): batch order %.3f | longest first %.3f | ideal 1.000" % (
    makespan(range(n)) / (n / 512 * dur.mean()), makespan(np.argsort(-dur)) / (n / 512 * dur.mean())))
# level-granular scheduling model: unit (pair, level) ~ c0 + c1*iters[level], units of a pair in sequence,
# any slot may continue any pair (the carried state is the 7-double pose)
stats = np.frombuffer(d["stats"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
it = stats["iters"][:, :4].astype(np.float64)
A = np.stack([np.full(n, 4.0), it.sum(axis=1)], axis=1)
c0, c1 = np.linalg.lstsq(A, dur, rcond=None)[0]
print("fit: pair cycles ~ 4*%.0f + %.0f*iterations (residual rms %.0f)" % (c0, c1, np.sqrt(np.mean((A @ [c0, c1] - dur) ** 2))))
def makespan_levels(slots=512):
    import heapq
    from collections import deque
    ready = deque((i, 3) for i in range(n))          # FIFO of ready units
    free = [(0.0, k) for k in range(slots)]            # (time the slot is free, slot)
    heapq.heapify(free)
    pending = []                                       # (finish time, pair, level) of running units
    t_end = 0.0
    while ready or pending:
        if ready and free and (not pending or free[0][0] <= pending[0][0]):
            t, k = heapq.heappop(free)
            i, l = ready.popleft()
            f = t + c0 + c1 * it[i, l]
            heapq.heappush(pending, (f, i, l, k))
        else:
            f, i, l, k = heapq.heappop(pending)
            t_end = max(t_end, f)
            heapq.heappush(free, (f, k))
            if l > 0: ready.append((i, l - 1))
            # a slot that is free earlier than f must not start the new unit before f
            free = [(max(tt, f) if not ready else tt, kk) for tt, kk in free]; heapq.heapify(free)
    return t_end
print("model makespan / ideal: whole pairs %.3f | level-granular FIFO %.3f" % (
    makespan(range(n)) / (n / 512 * dur.mean()), makespan_levels() / (n / 512 * dur.mean())))

# ---- the tail of a solo launch: which slots idle once the pair counter has run dry (device-wide 100 MHz clock)
t0, t1, slot = rt[:, 0], rt[:, 1], rt[:, 2].astype(int)
T0, T1 = t0.min(), t1.max()
span = T1 - T0
n_slots = 2 * 256
slots = np.unique(slot)
last_end = np.array([t1[slot == k].max() for k in slots])
first_begin = np.array([t0[slot == k].min() for k in slots])
busy = np.array([(t1[slot == k] - t0[slot == k]).sum() for k in slots])
dry = t0.max()                                        # the last claim: after it no slot finds a pair
idle_tail = (T1 - last_end).sum() + (n_slots - len(slots)) * span
print("solo launch, %d pairs on %d of %d slots: span %.1f us (first begin -> last end, 100 MHz clock)" % (a.pairs, len(slots), n_slots, span / 100.0))
print("  pairs per slot: " + " ".join("%d:%d" % (k, (np.bincount(slot)[slots] == k).sum()) for k in range(1, 6)))
print("  counter dry (last claim) at %.1f us = %.3f of the span" % ((dry - T0) / 100.0, (dry - T0) / span))
print("  slot-time idle after a slot's last pair: %.3f of slots x span (mean %.1f us per slot, max %.1f us)" % (
    idle_tail / (n_slots * span), (T1 - last_end).mean() / 100.0, (T1 - last_end).max() / 100.0))
print("  slot-time idle before a slot's first pair: %.4f; between pairs: %.4f" % (
    (first_begin - T0).sum() / (n_slots * span), ((last_end - first_begin) - busy).sum() / (n_slots * span)))
h, edges = np.histogram((T1 - last_end) / 100.0, bins=[0, 5, 10, 20, 30, 40, 60, 80, 120, 1e9])
print("  histogram of a slot's idle tail [us]: " + " ".join("%s:%d" % ("<%g" % edges[i + 1] if edges[i + 1] < 1e8 else ">=%g" % edges[i], h[i]) for i in range(len(h))))
# per-pair duration by whether its slot-mate was still running: pairs that END after the mate's last end ran part of their time alone
dur_rt = (t1 - t0) / 100.0
order = np.argsort(t0)
second = np.array([t0[i] > first_begin[np.searchsorted(slots, slot[i])] for i in range(a.pairs)])
print("  pair duration: first of a slot %.1f us, later pairs %.1f us (mean); p95 %.1f, max %.1f" % (
    dur_rt[~second].mean(), dur_rt[second].mean() if second.any() else float("nan"), np.percentile(dur_rt, 95), dur_rt.max()))
# CUs where both slots finished vs. one slot alone at the end
cu = slots // 2
alone = np.array([abs(last_end[cu == c][0] - last_end[cu == c][-1]) for c in np.unique(cu)]) / 100.0
print("  per CU, time one slot runs alone at the end: mean %.1f us, p95 %.1f us" % (alone.mean(), np.percentile(alone, 95)))
# arbitration by age: is the workgroup's second slot (its waves are the younger half) systematically slower than the first?
s0, s1 = dur_rt[slot % 2 == 0], dur_rt[slot % 2 == 1]
print("  pair duration by slot of the workgroup: slot 0 %.1f us (p95 %.1f), slot 1 %.1f us (p95 %.1f)" % (s0.mean(), np.percentile(s0, 95), s1.mean(), np.percentile(s1, 95)))
le0 = np.array([last_end[np.searchsorted(slots, k)] for k in slots if k % 2 == 0]); le1 = np.array([last_end[np.searchsorted(slots, k)] for k in slots if k % 2 == 1])
print("  a slot's last end after the launch's first begin: slot 0 %.1f us, slot 1 %.1f us (mean)" % ((le0 - T0).mean() / 100.0, (le1 - T0).mean() / 100.0))
pwv = pw[:, :5] / np.maximum(n_it, 1)[:, None]
print("  pass cycles per iteration by patch wave, slot 0: " + " ".join("%.0f" % v for v in pwv[slot % 2 == 0].mean(0)) + " | slot 1: " + " ".join("%.0f" % v for v in pwv[slot % 2 == 1].mean(0)))
