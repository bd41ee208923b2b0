#!/usr/bin/env python3
"""bench.py — frame-pair sparse alignments per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: ONE launch of
dsdtm_sparse_align_batch_device over `--pairs` independent 640x480 frame pairs per GPU
(BASELINE config 4's per-GPU share of config 2's shape: 4 pyramid levels, 300 patches,
cap 10 Gauss-Newton iterations), inputs resident in HBM when the timed region starts.
N GPUs = N processes (torch.distributed / RCCL only for the barrier + max-over-ranks timing;
the path itself needs no collective: independent pairs, weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     — algorithmic bytes (833,392 B per alignment) / kernel time
                 (HIP events on the launch stream) against 8 TB/s HBM
  cpu_baseline — the CPU oracle (line-faithful restatement of the reference) timed on this
                 box's host cores on the same pairs, plus the pose delta GPU vs CPU.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(width, height, levels, n_patches):
    """2 * sum_l (W/2^l)(H/2^l) [u8 ref+cur pyramids, each read once]
    + N*(2*4 + 3*8 + 3*8 + 1) [px, bearing, P_w, flag] + 2*96 [T_ref_w, T_cur_w in] + 96 + 4 [out]."""
    pyr = 0
    w, h = width, height
    for _ in range(levels):
        pyr += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    return 2 * pyr + n_patches * 57 + 292


def pmc_traffic(pairs_per_launch, b_alg):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/, collected with separate
    --pmc runs of this same command; bench.py cannot sample PMCs itself). Returns None when the summary
    is missing or was taken for another launch size."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_final.json")
    try:
        with open(path) as f:
            d = json.load(f)
        t = d["hbm_traffic_per_launch"]
        if int(t["algorithmic_bytes_per_launch"]) != pairs_per_launch * b_alg:
            return None
        return float(t["fetch_bytes_gfx950_corrected"]) + float(t["write_bytes"])
    except Exception:
        return None


def se3_exp_batch(xi):
    """numpy batch of SE(3) exponentials -> (n,4,4)."""
    from dsdtm_amd import synth
    return np.stack([synth.se3_exp(x) for x in xi])


def build_batch(torch, dev, ctx, cam, n_pairs, width, height, levels, n_patches, seed, stream):
    """Synthetic batch generated on the GPU (texture FFT + bicubic plane warp with torch, pyramid
    with the library's own pyrDown kernel). Returns a dict of device tensors + the BatchDesc."""
    import torch.nn.functional as F
    from dsdtm_amd import capi, synth

    ws, hs, strides, offs, pyr_bytes = capi.pyramid_layout(width, height, levels, 64)
    pitch = (pyr_bytes + 255) // 256 * 256
    ref_pyr = torch.zeros((n_pairs, pitch), dtype=torch.uint8, device=dev)
    cur_pyr = torch.zeros((n_pairs, pitch), dtype=torch.uint8, device=dev)

    rng = np.random.default_rng(seed)
    xi = np.concatenate([rng.uniform(-0.02, 0.02, (n_pairs, 3)), rng.uniform(-0.01, 0.01, (n_pairs, 3))], axis=1)
    depth = rng.uniform(1.0, 4.0, n_pairs)
    T_cr = se3_exp_batch(xi)                                                  # cur <- ref
    T_ref = np.stack([np.vstack([synth.random_pose(rng), [0, 0, 0, 1]]) for _ in range(n_pairs)])
    K = cam.K()
    Kinv = np.linalg.inv(K)
    n = np.array([0.0, 0.0, 1.0])
    Hrc = np.stack([np.linalg.inv(K @ (T_cr[i, :3, :3] + np.outer(T_cr[i, :3, 3], n) / depth[i]) @ Kinv)
                    for i in range(n_pairs)])                                  # cur px -> ref px

    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    fy = torch.fft.fftfreq(height, device=dev)[:, None]
    fx = torch.fft.rfftfreq(width, device=dev)[None, :]
    rad = torch.sqrt(fx * fx + fy * fy)
    rad[0, 0] = 1.0
    filt = rad.pow(-0.9) * torch.exp(-(rad / 0.25) ** 2)
    filt[0, 0] = 0.0
    uu, vv = torch.meshgrid(torch.arange(width, device=dev, dtype=torch.float32),
                            torch.arange(height, device=dev, dtype=torch.float32), indexing="xy")
    chunk = 32
    for i0 in range(0, n_pairs, chunk):
        i1 = min(n_pairs, i0 + chunk)
        white = torch.randn((i1 - i0, height, width), generator=g, device=dev)
        tex = torch.fft.irfft2(torch.fft.rfft2(white) * filt, s=(height, width))
        flat = tex.reshape(i1 - i0, -1)
        lo = torch.quantile(flat[:, ::16], 0.005, dim=1)[:, None, None]
        hi = torch.quantile(flat[:, ::16], 0.995, dim=1)[:, None, None]
        tex = ((tex - lo) / (hi - lo)).clamp(0, 1) * 255.0
        Hm = torch.tensor(Hrc[i0:i1], dtype=torch.float32, device=dev)
        den = Hm[:, 2, 0, None, None] * uu + Hm[:, 2, 1, None, None] * vv + Hm[:, 2, 2, None, None]
        xr = (Hm[:, 0, 0, None, None] * uu + Hm[:, 0, 1, None, None] * vv + Hm[:, 0, 2, None, None]) / den
        yr = (Hm[:, 1, 0, None, None] * uu + Hm[:, 1, 1, None, None] * vv + Hm[:, 1, 2, None, None]) / den
        grid = torch.stack([xr / (width - 1) * 2 - 1, yr / (height - 1) * 2 - 1], dim=-1)
        cur = F.grid_sample(tex[:, None], grid, mode="bicubic", padding_mode="reflection", align_corners=True)[:, 0]
        ref_pyr[i0:i1, :width * height] = tex.round().clamp(0, 255).to(torch.uint8).reshape(i1 - i0, -1)
        cur_pyr[i0:i1, :width * height] = cur.round().clamp(0, 255).to(torch.uint8).reshape(i1 - i0, -1)
        del white, tex, cur, grid
    torch.cuda.synchronize()

    # pyramid levels 1.. with the product's pyrDown kernel (bit-exact cv::pyrDown; tests/test_align2d_gpu.py)
    wa, ha, sa = (C.c_int * levels)(*ws), (C.c_int * levels)(*hs), (C.c_int * levels)(*strides)
    oa = (C.c_size_t * levels)(*offs)
    for t in (ref_pyr, cur_pyr):
        ctx.check(ctx.lib.dsdtm_pyrdown_batch_device(ctx.handle, t.data_ptr(), pitch, n_pairs, levels, wa, ha, sa, oa,
                                                     stream.cuda_stream))
    stream.synchronize()

    # features: Tracking's view of the reference frame (px f32, unit bearing, world point)
    px = np.stack([rng.uniform(30, width - 30, (n_pairs, n_patches)),
                   rng.uniform(30, height - 30, (n_pairs, n_patches))], axis=2).astype(np.float32)
    if os.environ.get("DSDTM_BENCH_CLUSTER"):   # diagnostic (tools/stamps.py): all features of a pair inside a WxH window
        cw, ch = (int(v) for v in os.environ["DSDTM_BENCH_CLUSTER"].split("x"))
        px = np.stack([rng.uniform(width / 2 - cw / 2, width / 2 + cw / 2, (n_pairs, n_patches)),
                       rng.uniform(height / 2 - ch / 2, height / 2 + ch / 2, (n_pairs, n_patches))], axis=2).astype(np.float32)
    if os.environ.get("DSDTM_BENCH_SORT"):      # diagnostic: spatially coherent feature order (128-px strips, then rows)
        key = (px[:, :, 0] // 128).astype(np.int64) * 100000 + px[:, :, 1].astype(np.int64)
        order = np.argsort(key, axis=1, kind="stable")
        px = np.take_along_axis(px, order[:, :, None], axis=1)
    bearing = synth.bearing_from_px(cam, px.reshape(-1, 2)).reshape(n_pairs, n_patches, 3)
    X_r = bearing * (depth[:, None, None] / bearing[:, :, 2:3])
    Rr, tr = T_ref[:, :3, :3], T_ref[:, :3, 3]
    p_world = np.einsum("nji,npj->npi", Rr, X_r - tr[:, None, :])             # R^T (X - t)
    initial = np.ones((n_pairs, n_patches), np.uint8)
    T_true = np.einsum("nij,njk->nik", T_cr, T_ref)[:, :3, :]

    d = dict(
        ref_pyr=ref_pyr, cur_pyr=cur_pyr,
        px=torch.from_numpy(px).to(dev), bearing=torch.from_numpy(bearing).to(dev),
        p_world=torch.from_numpy(p_world).to(dev), initial=torch.from_numpy(initial).to(dev),
        T_ref_w=torch.from_numpy(np.ascontiguousarray(T_ref[:, :3, :].reshape(n_pairs, 12))).to(dev),
        T_seed=torch.from_numpy(np.ascontiguousarray(T_ref[:, :3, :].reshape(n_pairs, 12))).to(dev),
        T_cur_w=torch.zeros((n_pairs, 12), dtype=torch.float64, device=dev),
        n_tracked=torch.zeros(n_pairs, dtype=torch.int32, device=dev),
        stats=torch.zeros((n_pairs, capi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev),
        T_true=T_true, pitch=pitch,
    )
    b = capi.BatchDesc()
    b.n_pairs, b.max_features, b.levels = n_pairs, n_patches, levels
    for l in range(levels):
        b.width[l], b.height[l], b.stride[l], b.level_offset[l] = ws[l], hs[l], strides[l], offs[l]
    b.pyr_pitch = pitch
    b.ref_pyr, b.cur_pyr = d["ref_pyr"].data_ptr(), d["cur_pyr"].data_ptr()
    b.px_xy, b.bearing, b.p_world = d["px"].data_ptr(), d["bearing"].data_ptr(), d["p_world"].data_ptr()
    b.initial, b.n_features = d["initial"].data_ptr(), None
    b.T_ref_w, b.T_cur_w = d["T_ref_w"].data_ptr(), d["T_cur_w"].data_ptr()
    b.n_tracked, b.stats = d["n_tracked"].data_ptr(), d["stats"].data_ptr()
    d["desc"] = b
    return d


def cpu_baseline(d, cam_struct, prm, sample, threads):
    """Times the CPU oracle on the first `sample` pairs (copied to the host) and returns
    (alignments/s, poses (sample,12), n_tracked, stats array)."""
    from dsdtm_amd import capi
    from tests import oracle_lib
    lib = oracle_lib.load()
    host = {k: d[k][:sample].cpu().numpy().copy() for k in ("ref_pyr", "cur_pyr", "px", "bearing", "p_world", "initial", "T_ref_w")}
    T = d["T_seed"][:sample].cpu().numpy().copy()
    nt = np.zeros(sample, np.int32)
    st = np.zeros(sample, capi.STATS_DTYPE)
    src = d["desc"]
    b = capi.BatchDesc()
    C.memmove(C.byref(b), C.byref(src), C.sizeof(b))
    b.n_pairs = sample
    b.ref_pyr, b.cur_pyr = host["ref_pyr"].ctypes.data, host["cur_pyr"].ctypes.data
    b.px_xy, b.bearing, b.p_world = host["px"].ctypes.data, host["bearing"].ctypes.data, host["p_world"].ctypes.data
    b.initial, b.n_features = host["initial"].ctypes.data, None
    b.T_ref_w, b.T_cur_w = host["T_ref_w"].ctypes.data, T.ctypes.data
    b.n_tracked, b.stats = nt.ctypes.data, st.ctypes.data
    secs = lib.oracle_sparse_align_batch_timed(C.byref(b), C.byref(cam_struct), C.byref(prm), threads)
    return sample / secs, T, nt, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the steady state (clocks and launch pipeline settle after ~20 launches:
    # 20 steps behind 3 warm-up launches measure 3.75 M alignments/s, >= 200 steps 4.06 M); 0.13 s of GPU time
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--pairs", type=int, default=1024, help="frame pairs per GPU per step")
    ap.add_argument("--patches", type=int, default=300)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="pairs timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    from dsdtm_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if not os.path.exists(capi.lib_path()):   # tooling convenience only: the product itself never builds or falls back
        from dsdtm_amd.csrc import build as hip_build
        if rank == 0:
            hip_build.build(verbose=False)
        if world > 1:
            dist.barrier()
    ctx = capi.Context(local_rank)          # fails loudly without the HIP library / a gfx950 device
    cam = synth.Camera.tum(args.width, args.height)
    cam_struct = capi.camera_struct(cam)
    prm = capi.AlignParams(args.levels, 0, args.iters, 15)
    stream = torch.cuda.Stream(device=dev)

    from dsdtm_amd import shard
    d = build_batch(torch, dev, ctx, cam, args.pairs, args.width, args.height, args.levels, args.patches,
                    seed=shard.batch_seed(0xD5D7, rank), stream=stream)
    desc = d["desc"]
    ws_bytes = ctx.lib.dsdtm_sparse_align_workspace_bytes(C.byref(desc))
    if ws_bytes:
        ctx.check(ctx.lib.dsdtm_reserve(ctx.handle, ws_bytes))

    # The pose argument is in/out (seed in, result out: Tracking seeds cur.pose = last.pose, src/Tracking.cpp:201).
    # Every step gets its own copy of the seed poses, resident in HBM before the timed region like the rest
    # of the input, so that a step is exactly one launch of the hot path (no re-seeding copy between steps).
    n_slots = args.warmup + args.steps
    d["T_steps"] = d["T_seed"].unsqueeze(0).repeat(n_slots, 1, 1).contiguous()
    descs = []
    for k in range(n_slots):
        dk = capi.BatchDesc.from_buffer_copy(bytes(desc))
        dk.T_cur_w = d["T_steps"][k].data_ptr()
        descs.append(dk)

    def step(k):
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(descs[k]), C.byref(cam_struct), C.byref(prm),
                                                          stream.cuda_stream))

    def barrier():
        if world > 1:
            dist.barrier()

    for k in range(args.warmup):
        step(k)
    stream.synchronize()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)           # HIP events on the stream the kernel is launched on
        step(args.warmup + k)
        ev[k][1].record(stream)
    stream.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = shard.max_over_ranks(elapsed, dist, dev)
    d["T_cur_w"] = d["T_steps"][n_slots - 1]          # the last step's results are the ones checked below
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    if rank == 0:
        n_total = args.pairs * world * args.steps
        value = n_total / elapsed
        b_alg = algorithmic_bytes(args.width, args.height, args.levels, args.patches)
        k_avg = float(np.mean(kernel_ms))
        achieved = args.pairs * b_alg / (k_avg * 1e-3) / 1e9
        stats = np.frombuffer(d["stats"].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
        iters = stats["iters"][:, :args.levels]
        Tg = d["T_cur_w"].cpu().numpy()
        ntg = d["n_tracked"].cpu().numpy()
        err = np.array([synth.pose_error(Tg[i], d["T_true"][i]) for i in range(args.pairs)])
        out = {
            "metric": "frame-pair alignments/sec (640x480, 4 lvls, ~300 patches, 10 GN iters); pose delta vs CPU",
            "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4 per-GPU share of config 2's shape: {args.pairs} independent "
                                   f"{args.width}x{args.height} frame pairs per GPU per step, {args.levels} pyramid "
                                   f"levels, {args.patches} patches, cap {args.iters} GN iterations, one launch per step",
                       "pairs_per_gpu": args.pairs, "patches": args.patches, "levels": args.levels,
                       "max_iters": args.iters, "parallelism": f"independent pairs x{world} (no collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.pairs, b_alg),
                         "traffic_note": "bytes per launch, rocprofv3 FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE "
                                         "from profiles/r01_bench_pmc_final.json; algorithmic bytes per launch = "
                                         + str(args.pairs * b_alg),
                         "kernel": "sparse_align_reg_kernel", "kernel_ms_avg": k_avg,
                         "kernel_ms_min": float(np.min(kernel_ms)),
                         "algorithmic_bytes_per_alignment": b_alg, "alignments_per_launch": args.pairs},
            "executed_iterations_per_level_mean": [float(x) for x in iters.mean(axis=0)],
            "executed_iterations_total_mean": float(iters.sum(axis=1).mean()),
            "n_tracked_mean": float(ntg.mean()),
            "err_vs_ground_truth_median": {"rad": float(np.median(err[:, 0])), "m": float(np.median(err[:, 1]))},
            "library": ctx.lib.dsdtm_version().decode(),
        }
        if not args.no_cpu and world == 1:
            sample = min(args.cpu_sample, args.pairs)
            cpu_rate, To, nto, sto = cpu_baseline(d, cam_struct, prm, sample, 1)
            dl = np.array([synth.pose_error(Tg[i], To[i]) for i in range(sample)])
            out["cpu_baseline"] = {
                "value": cpu_rate, "unit": "alignments/s", "cores": 1, "kind": "port",
                "sample": f"first {sample} pairs of the GPU batch (same bytes), CPU oracle (C restatement of the "
                          f"reference, -O3 -msse..-mssse3 as reference CMakeLists.txt:5-8), one thread as the "
                          f"reference's tracking thread",
                "host_cores_available": os.cpu_count(),
            }
            out["pose_delta_vs_cpu"] = {"max_rad": float(dl[:, 0].max()), "max_m": float(dl[:, 1].max()),
                                        "pairs_checked": int(sample),
                                        "n_tracked_equal": bool(np.array_equal(ntg[:sample], nto)),
                                        "iterations_equal": bool(np.array_equal(stats["iters"][:sample], sto["iters"])),
                                        "tolerance": "1e-4 rad / 1e-4 m (north_star)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
