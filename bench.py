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

import bench_line  # noqa: E402

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


PMC_SUMMARY = os.path.join("profiles", "r06_bench_pmc.json")


def library_sha():
    """sha256 (first 16 hex digits) of the libdsdtm_amd.so this process loads: the counter summary under profiles/
    carries the hash of the binary it was taken with, and its numbers are only reported for that binary."""
    import hashlib
    from dsdtm_amd import capi
    try:
        with open(capi.lib_path(), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def library_source_sha():
    """sha256 over the sources + flags the in-tree library is built from (dsdtm_amd/csrc/build.py: source_sha), valid as
    an identity of the LOADED library only when that library is up to date with them."""
    try:
        from dsdtm_amd.csrc import build as hip_build
        return None if hip_build.needs_build() else hip_build.source_sha()
    except Exception:
        return None


def pmc_summary(path=PMC_SUMMARY):
    """The committed counter summary, or {} when it is missing or describes another build of the library than the loaded
    one: the binary's hash must match, or — hipcc's output is not reproducible byte for byte — the hash of the sources and
    flags it was built from, with the loaded library up to date with those sources."""
    try:
        with open(os.path.join(ROOT, path)) as f:
            d = json.load(f)
    except Exception:
        return {}
    if d.get("profile_binary_sha") and d.get("profile_binary_sha") == library_sha():
        return d
    if d.get("profile_source_sha") and d.get("profile_source_sha") == library_source_sha():
        return d
    return {}


def pmc_traffic(kernel_substr, algorithmic_bytes_per_launch, path=PMC_SUMMARY):
    """HBM bytes per launch of the kernel whose name contains `kernel_substr`, from the committed rocprofv3
    PMC passes (profiles/, separate --pmc runs of this same command, folded by tools/summarize_profile.py;
    bench.py cannot sample PMCs itself): FETCH_SIZE [KiB] x 1024 x 2 (gfx950 correction of the microarch
    guide) + WRITE_SIZE [KiB] x 1024. None when the summary is missing, was taken for another launch size, or
    was taken with another build of the library than the one loaded now (profile_binary_sha)."""
    try:
        d = pmc_summary(path)
        for t in d.get("hbm_traffic_per_launch", []):
            if kernel_substr in t["kernel"] and int(t["algorithmic_bytes_per_launch"]) == int(algorithmic_bytes_per_launch):
                return float(t["fetch_bytes_gfx950_corrected"]) + float(t["write_bytes"])
    except Exception:
        pass
    return None


FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X FP64 vector: half the 157.3 TFLOP/s FP32 vector rate of MI355X_MICROARCH.md


def pmc_fp64_flops(kernel_substr, case="solo", path=PMC_SUMMARY):
    """FP64 flops per launch of the BASELINE workload (1024 pairs) from the committed SQ_INSTS_VALU_*_F64 pass
    (64 x (ADD + MUL + 2 FMA + TRANS) wave instructions; tools/profile.sh, tools/summarize_profile.py). None if absent."""
    try:
        d = pmc_summary(path)
        for t in d.get("fp64_per_launch", []):
            if t["case"] == case and kernel_substr in t["kernel"]:
                return float(t["fp64_flops_per_launch"])
    except Exception:
        pass
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
        if os.environ["DSDTM_BENCH_SORT"] == "row":
            key = px[:, :, 1].astype(np.int64) * 4096 + px[:, :, 0].astype(np.int64)
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


def usable_cpus():
    """Host threads this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(parts[0]) / int(parts[1])))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(math.ceil(q / int(f.read())))))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class HostBatch:
    """The first `sample` pairs of a device batch copied to the host, as a dsdtm_batch_desc the CPU oracle takes."""

    def __init__(self, d, sample):
        from dsdtm_amd import capi
        self.sample = sample
        self.host = {k: d[k][:sample].cpu().numpy().copy() for k in
                     ("ref_pyr", "cur_pyr", "px", "bearing", "p_world", "initial", "T_ref_w")}
        self.seed = d["T_seed"][:sample].cpu().numpy().copy()
        self.T = self.seed.copy()
        self.nt = np.zeros(sample, np.int32)
        self.st = np.zeros(sample, capi.STATS_DTYPE)
        b = capi.BatchDesc()
        C.memmove(C.byref(b), C.byref(d["desc"]), C.sizeof(b))
        h = self.host
        b.n_pairs = sample
        b.ref_pyr, b.cur_pyr = h["ref_pyr"].ctypes.data, h["cur_pyr"].ctypes.data
        b.px_xy, b.bearing, b.p_world = h["px"].ctypes.data, h["bearing"].ctypes.data, h["p_world"].ctypes.data
        b.initial, b.n_features = h["initial"].ctypes.data, None
        b.T_ref_w, b.T_cur_w = h["T_ref_w"].ctypes.data, self.T.ctypes.data
        b.n_tracked, b.stats = self.nt.ctypes.data, self.st.ctypes.data
        self.desc = b

    def run(self, lib, cam_struct, prm, threads, min_seconds=0.0, max_reps=64):
        """Times the oracle over the sample (repeated until `min_seconds` of wall time have been measured);
        returns alignments/s of the best repetition-independent average: total pairs / total seconds."""
        total, reps = 0.0, 0
        while reps < max_reps and (reps == 0 or total < min_seconds):
            self.T[...] = self.seed
            total += lib.oracle_sparse_align_batch_timed(C.byref(self.desc), C.byref(cam_struct), C.byref(prm), threads)
            reps += 1
        return self.sample * reps / total, reps


def cpu_baselines(d, cam_struct, prm, sample):
    """CPU legs of the bench line: the C restatement of the reference timed on this box's host
    cores over the first `sample` pairs of the GPU batch — one thread with the reference's ISA flags (the
    reference runs this path on its tracking thread), one thread with -march=native, and all usable host
    threads with one alignment per thread. Returns (dict of JSON objects, poses, n_tracked, stats)."""
    from tests import oracle_lib
    hb = HostBatch(d, sample)
    cores = usable_cpus()
    out = {}
    lib = oracle_lib.load()
    rate1, reps1 = hb.run(lib, cam_struct, prm, 1, min_seconds=4.0)
    T1, nt1, st1 = hb.T.copy(), hb.nt.copy(), hb.st.copy()
    out["cpu_baseline"] = {
        "value": rate1, "unit": "alignments/s", "cores": 1, "kind": "port",
        "sample": f"first {sample} pairs of the GPU batch (same bytes) x {reps1} repetitions, CPU oracle (C restatement "
                  f"of the reference, gcc -O3 -msse..-mssse3 as reference CMakeLists.txt:5-8), one thread as the "
                  f"reference's tracking thread",
        "host_cpu": cpu_model(), "host_threads_usable": cores, "host_threads_total": os.cpu_count()}
    rate_all, reps_all = hb.run(lib, cam_struct, prm, cores, min_seconds=4.0)
    out["cpu_baseline_all_cores"] = {
        "value": rate_all, "unit": "alignments/s", "cores": cores, "kind": "port",
        "sample": f"the same {sample} pairs x {reps_all} repetitions, one alignment per OpenMP thread (dynamic schedule), "
                  f"{cores} threads = every host thread this process may use (affinity mask / cgroup quota)"}
    try:
        libn = oracle_lib.load(native=True)      # built on THIS host (file name carries the CPU tag)
        rate_n, reps_n = hb.run(libn, cam_struct, prm, 1, min_seconds=3.0)
        out["cpu_baseline_native"] = {
            "value": rate_n, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"the same {sample} pairs x {reps_n} repetitions, one thread, gcc -O3 -march=native "
                      f"(as ROS_Demo/CMakeLists.txt of the reference)"}
    except Exception as e:                        # no compiler on the box: the row is reported as missing, not faked
        out["cpu_baseline_native"] = {"value": None, "error": str(e)[:200]}
    return out, T1, nt1, st1


def rank_parity(d, cam_struct, prm, n_pairs, threads):
    """This rank's own parity leg (every rank of an N-GPU run, after the timed region): the first `n_pairs` pairs of ITS batch
    through the CPU oracle — the checker, never timed into `value` — against the poses / counts / iterations its last step
    left on the device. Returns [max_rad, max_m, n_tracked_equal, iterations_equal, pairs_checked] as floats (gathered over
    the ranks by the caller). The reference's result IS the pose (src/Sprase_ImageAlign.cpp:57-59)."""
    from dsdtm_amd import capi, synth
    from tests import oracle_lib
    hb = HostBatch(d, n_pairs)
    hb.run(oracle_lib.load(), cam_struct, prm, max(1, threads))
    Tg = d["T_cur_w"][:n_pairs].cpu().numpy()
    ntg = d["n_tracked"][:n_pairs].cpu().numpy()
    stg = np.frombuffer(d["stats"][:n_pairs].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
    dl = np.array([synth.pose_error(Tg[i], hb.T[i]) for i in range(n_pairs)])
    bad = not np.isfinite(dl).all()
    return [float("inf") if bad else float(dl[:, 0].max()), float("inf") if bad else float(dl[:, 1].max()),
            float(np.array_equal(ntg, hb.nt)), float(np.array_equal(stg["iters"], hb.st["iters"])), float(n_pairs)]


def aggregate_rank_checks(rank_checks):
    """The gathered rank_parity rows (one per rank) -> (pose_delta_vs_cpu object, passed). A rank that checked nothing, a
    non-finite delta, any delta over north_star's 1e-4 rad / 1e-4 m or a differing n_tracked fails the whole job."""
    mr, mm = max(r[0] for r in rank_checks), max(r[1] for r in rank_checks)
    pd = {"max_rad": mr, "max_m": mm, "pairs_checked": int(sum(r[4] for r in rank_checks)),
          "n_tracked_equal": bool(all(r[2] == 1.0 for r in rank_checks)),
          "iterations_equal": bool(all(r[3] == 1.0 for r in rank_checks)),
          "ranks_checked": sum(1 for r in rank_checks if r[4] > 0),
          "per_rank_max_rad": [r[0] for r in rank_checks], "per_rank_max_m": [r[1] for r in rank_checks],
          "tolerance": "1e-4 rad / 1e-4 m (north_star)"}
    ok = bool(mr <= 1e-4 and mm <= 1e-4 and pd["n_tracked_equal"] and pd["ranks_checked"] == len(rank_checks))
    return pd, ok


def hip_event_ms(torch, stream, fn, launches):
    """Average duration of `launches` calls of fn(stream), HIP events on `stream` around each call."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in ev:
        a.record(stream)
        fn(stream)
        b.record(stream)
    stream.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return float(np.mean(t)), float(np.min(t))


def roofline_block(kernel, alg_bytes_per_launch, ms_avg, ms_min, units_per_launch, alg_bytes_per_unit, unit_name, extra=None):
    achieved = alg_bytes_per_launch / (ms_avg * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": pmc_traffic(kernel, alg_bytes_per_launch),
         "traffic_note": "HBM bytes per launch: rocprofv3 FETCH_SIZE [KiB] x 1024 x 2 (gfx950 correction) + WRITE_SIZE "
                         "[KiB] x 1024, separate --pmc passes of this command (" + PMC_SUMMARY + "); null when no "
                         "summary was committed for this launch size; algorithmic bytes per launch = "
                         + str(int(alg_bytes_per_launch)),
         "kernel": kernel, "kernel_ms_avg": ms_avg,
         "algorithmic_bytes_per_" + unit_name: alg_bytes_per_unit, unit_name + "s_per_launch": units_per_launch}
    if ms_min is not None:
        r["kernel_ms_min"] = ms_min
    if r["traffic"] is None:
        r["traffic_note"] += "; no committed counter summary matches this launch size AND the loaded binary (sha " + str(library_sha()) + ")"
        r["hbm_frac_measured"] = None
    else:
        # the fraction of the HBM peak the kernel's REAL traffic amounts to (frac prices the algorithmic bytes)
        r["hbm_frac_measured"] = r["traffic"] / (ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS
    if extra:
        r.update(extra)
    return r


def secondary_entries(torch, dev, ctx, cam_struct_640, stream, args):
    """Other shapes of the path, each with its own roofline block, rank 0 at N=1 only, outside
    the timed region: the batched alignment for BASELINE configs 3 and 5 (1000 patches at 640x480; 2000 patches
    at 1280x960) and the pyramid kernel. Every alignment entry is checked against the CPU oracle on a sample."""
    from dsdtm_amd import capi, synth
    from tests import oracle_lib
    out = []
    lib = oracle_lib.load()
    for name, width, height, n_pairs, n_patches, launches, levels, iters in (
            ("config 3 shape: 640x480, 4 levels, 1000 patches, cap 10", 640, 480, 1024, 1000, 30, args.levels, args.iters),
            ("config 5 shape: 1280x960, 4 levels, 2000 patches, cap 10", 1280, 960, 256, 2000, 30, args.levels, args.iters),
            # what DSDTM's Tracking really constructs (src/Tracking.cpp:20-24,37; Config/default.yaml:64-65,93): 5 levels, 8 iterations,
            # Camera.Max_tkfts = 200 features per frame
            ("Tracking's own arguments: 640x480, 5 levels, 190 patches, cap 8", 640, 480, 1024, 190, 30, 5, 8)):
        cam = synth.Camera.tum(width, height)
        cs = capi.camera_struct(cam)
        prm = capi.AlignParams(levels, 0, iters, 15)
        d = build_batch(torch, dev, ctx, cam, n_pairs, width, height, levels, n_patches, seed=0xC0DE + n_patches, stream=stream)
        desc = d["desc"]

        def launch(s, d=d, desc=desc, cs=cs, prm=prm):
            d["T_cur_w"].copy_(d["T_seed"])
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(desc), C.byref(cs), C.byref(prm), s.cuda_stream))

        def launch_timed(s, d=d, desc=desc, cs=cs, prm=prm):
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(desc), C.byref(cs), C.byref(prm), s.cuda_stream))

        with torch.cuda.stream(stream):
            for _ in range(8):                # (round 5: 5 timed launches behind 3 warm-ups read 5-8 % slower than the profiler's 346)
                launch(stream)
            # timed launches restart from the seed poses: the re-seeding copy sits between the event pairs
            ev = []
            for _ in range(launches):
                d["T_cur_w"].copy_(d["T_seed"])
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                launch_timed(stream)
                b.record(stream)
                ev.append((a, b))
        stream.synchronize()
        ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, stream.cuda_stream))
        t = [a.elapsed_time(b) for a, b in ev]
        ms_avg, ms_min = float(np.mean(t)), float(np.min(t))
        b_alg = algorithmic_bytes(width, height, levels, n_patches)
        # the same launches issued on four streams in turn, as the main line issues its steps (each launch in flight has
        # its own pose / count buffers; the statistics buffer is shared and not read here)
        ns, nl = 4, 24
        sts = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        bufs = [(d["T_seed"].clone(), torch.zeros_like(d["n_tracked"])) for _ in range(nl)]
        dks = []
        for Tb, nb in bufs:
            dk = capi.BatchDesc.from_buffer_copy(bytes(desc))
            dk.T_cur_w, dk.n_tracked, dk.stats = Tb.data_ptr(), nb.data_ptr(), None
            dks.append(dk)
        torch.cuda.synchronize()
        for k in range(ns):                                  # per-stream scratch and counters are set up by a first launch
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(dks[k]), C.byref(cs), C.byref(prm), sts[k].cuda_stream))
        torch.cuda.synchronize()
        for Tb, _ in bufs:
            Tb.copy_(d["T_seed"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nl):
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(dks[k]), C.byref(cs), C.byref(prm), sts[k % ns].cuda_stream))
        torch.cuda.synchronize()
        span_ms = (time.perf_counter() - t0) * 1e3 / nl
        for s_ in sts:
            ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, s_.cuda_stream))
        same = all(torch.equal(Tb, bufs[0][0]) for Tb, _ in bufs) and torch.equal(bufs[0][0], d["T_cur_w"])
        sample = 32
        hb = HostBatch(d, sample)
        cpu_rate, _ = hb.run(lib, cs, prm, usable_cpus())
        Tg = d["T_cur_w"][:sample].cpu().numpy()
        dl = np.array([synth.pose_error(Tg[i], hb.T[i]) for i in range(sample)])
        stats = np.frombuffer(d["stats"][:sample].cpu().numpy().tobytes(), dtype=capi.STATS_DTYPE)
        out.append({
            "key": f"align_{n_pairs}x{n_patches}_{width}x{height}" + ("" if (levels, iters) == (args.levels, args.iters) else f"_L{levels}_cap{iters}"),
            "workload": f"{n_pairs} independent pairs per launch, {name}",
            "value": n_pairs / (ms_avg * 1e-3), "unit": "alignments/s",
            "value_note": "one launch at a time (HIP events around every launch)",
            "value_four_streams": n_pairs / (span_ms * 1e-3),
            "four_streams_note": f"{nl} launches on {ns} streams in turn, wall time / launches = {span_ms:.4f} ms per launch; results of all "
                                 f"launches bit-identical to the one-stream launch: {bool(same)}",
            "roofline": roofline_block(d.get("kernel_name", "sparse_align"), n_pairs * b_alg, ms_avg, ms_min, n_pairs, b_alg, "alignment"),
            "pose_delta_vs_cpu": {"max_rad": float(dl[:, 0].max()), "max_m": float(dl[:, 1].max()), "pairs_checked": sample,
                                  "n_tracked_equal": bool(np.array_equal(d["n_tracked"][:sample].cpu().numpy(), hb.nt)),
                                  "iterations_equal": bool(np.array_equal(stats["iters"], hb.st["iters"]))},
            "cpu_all_cores_alignments_per_s": cpu_rate})
        del d
        torch.cuda.empty_cache()

    # Frame::ComputeImagePyramid: levels 1..3 of 2048 640x480 pyramids per launch group
    n_img, width, height, levels = 2048, 640, 480, args.levels
    ws, hs, strides, offs, pyr_bytes = capi.pyramid_layout(width, height, levels, 64)
    pitch = (pyr_bytes + 255) // 256 * 256
    pyr = torch.randint(0, 256, (n_img, pitch), dtype=torch.uint8, device=dev)
    wa, ha, sa = (C.c_int * levels)(*ws), (C.c_int * levels)(*hs), (C.c_int * levels)(*strides)
    oa = (C.c_size_t * levels)(*offs)

    def pyr_launch(s):
        ctx.check(ctx.lib.dsdtm_pyrdown_batch_device(ctx.handle, pyr.data_ptr(), pitch, n_img, levels, wa, ha, sa, oa, s.cuda_stream))

    for _ in range(3):
        pyr_launch(stream)
    ms_avg, ms_min = hip_event_ms(torch, stream, pyr_launch, 20)
    b_pyr = sum(ws[l] * hs[l] + ws[l + 1] * hs[l + 1] for l in range(levels - 1))     # level l read once, level l+1 written once
    out.append({
        "key": "pyramid", "workload": f"Frame::ComputeImagePyramid: levels 1..{levels - 1} of {n_img} {width}x{height} pyramids per call",
        "value": n_img / (ms_avg * 1e-3), "unit": "pyramids/s",
        "roofline": roofline_block("pyrdown", n_img * b_pyr, ms_avg, ms_min, n_img, b_pyr, "pyramid")})
    del pyr
    torch.cuda.empty_cache()
    out += tracked_frame_entries(torch, dev, ctx, stream)
    torch.cuda.empty_cache()
    # what Tracking sees: one pair / one frame at a time (bench_tracking.py)
    import bench_tracking
    out += bench_tracking.single_pair_entries(torch, dev, ctx, stream)
    out.append(bench_tracking.tracked_frame_entry(torch, dev, ctx, stream))
    torch.cuda.empty_cache()
    out.append(streamed_entry(torch, dev, ctx, synth.Camera.tum(args.width, args.height), cam_struct_640, args))
    return out


def tracked_frame_entries(torch, dev, ctx, stream):
    """The other kernels of a tracked frame, each as a device-resident batch with its own roofline block: Align2D
    (src/Feature_alignment.cpp:318-417), the pose-only refinement (src/Optimizer.cpp:20-101) and the detector's image
    work (src/Feature_detection.cpp:69-154), and FindMatchDirect = warp prelude + Align2D for the candidates of many current
    frames in one call (src/Feature_alignment.cpp:128-275). Device time by HIP events on the launch stream."""
    from dsdtm_amd import capi, synth
    from tests import helpers
    out = []

    def timed(fn, reps=20, warm=3):
        for _ in range(warm):
            fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / reps

    # ---- Align2D: 262144 features on one 1280x960 3-level pyramid (config 5's refinement step, batched)
    Wa, Ha, La, M, base = 1280, 960, 3, 262144, 4096
    tex = np.clip(np.rint(synth.make_texture(Ha, Wa, 5)), 0, 255).astype(np.uint8)
    pyr = synth.build_pyramid(tex, La)
    ws, hs, ss, offs, nb = capi.pyramid_layout(Wa, Ha, La)
    packed = np.zeros(nb, np.uint8)
    for l in range(La):
        packed[offs[l]:offs[l] + ws[l] * hs[l]] = pyr[l].reshape(-1)
    rng = np.random.default_rng(0)
    cs = np.stack([rng.uniform(20, Wa - 20, base), rng.uniform(20, Ha - 20, base)], 1)
    pb, p = helpers.make_border_patches(pyr[0], cs)
    rep_ = M // base
    d_pyr = torch.from_numpy(packed).to(dev)
    d_pb, d_p = torch.from_numpy(np.tile(pb, (rep_, 1))).to(dev), torch.from_numpy(np.tile(p, (rep_, 1))).to(dev)
    d_px0 = torch.from_numpy(np.tile(cs, (rep_, 1)) + rng.uniform(-1.5, 1.5, (M, 2))).to(dev)
    d_px = d_px0.clone()
    d_lv = torch.zeros(M, dtype=torch.int32, device=dev)
    d_cv = torch.zeros(M, dtype=torch.uint8, device=dev)
    img = capi.ImageDesc()
    img.levels = La
    for l in range(La):
        img.width[l], img.height[l], img.stride[l], img.level_offset[l] = ws[l], hs[l], ss[l], offs[l]
    img.bytes, img.data = nb, d_pyr.data_ptr()

    def a2d():
        with torch.cuda.stream(stream):
            d_px.copy_(d_px0, non_blocking=True)
        ctx.check(ctx.lib.dsdtm_align2d_batch_device(ctx.handle, C.byref(img), d_pb.data_ptr(), d_p.data_ptr(), d_lv.data_ptr(), d_px.data_ptr(),
                                                     d_cv.data_ptr(), 10, M, stream.cuda_stream))
    ms = timed(a2d)
    b_feat = 100 + 64 + 16 + 17                      # bordered patch + patch + pixel in/out + level and flag
    alg = M * b_feat + ws[0] * hs[0]                 # + the level image the features sit on, once
    out.append({"key": "align2d", "workload": f"Feature_Alignment::Align2DGaussNewton: {M} features per call on one {Wa}x{Ha} level, cap 10 iterations "
                            f"(four features per wavefront, float sums in the reference's order)",
                "value": M / (ms * 1e-3), "unit": "features/s", "converged_fraction": float(d_cv.float().mean().item()),
                "roofline": roofline_block("align2d", alg, ms, None, M, b_feat, "feature",
                                           {"note": "bound by VALU issue and the latency of <= 10 dependent iterations (64 bilinear samples, then 64 "
                                                    "sequential float subtractions per feature); the bytes are 197 per feature + the level image once"})})
    del d_pb, d_p, d_px0, d_px, d_lv, d_cv

    # ---- pose-only refinement: 4096 frames x 200 features (observations on levels 0..3)
    F_, N_, nb_ = 4096, 200, 64
    probs = [synth.make_pose_problem(1000 + k, n=N_, max_level=3) for k in range(nb_)]
    stack = lambda f: np.concatenate([np.stack([f(q) for q in probs])] * (F_ // nb_))
    tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_b, d_w, d_l, d_u = tdev(stack(lambda q: q.bearing)), tdev(stack(lambda q: q.p_world)), tdev(stack(lambda q: q.level)), tdev(stack(lambda q: q.use))
    d_T0 = tdev(stack(lambda q: q.T_seed.reshape(12)))
    d_T = d_T0.clone()
    d_rn = torch.zeros((F_, N_), dtype=torch.float64, device=dev)
    d_sm = torch.zeros((F_, C.sizeof(capi.PoseOptSummary)), dtype=torch.uint8, device=dev)
    pp = capi.PoseOptParams(100, 0)
    fpo = ctx.lib.dsdtm_pose_optimization_batch_device
    fpo.restype = C.c_int
    fpo.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(capi.PoseOptParams), C.c_void_p, C.c_void_p, C.c_void_p]

    def po():
        with torch.cuda.stream(stream):
            d_T.copy_(d_T0, non_blocking=True)
        ctx.check(fpo(ctx.handle, F_, N_, None, d_b.data_ptr(), d_w.data_ptr(), d_l.data_ptr(), d_u.data_ptr(), d_T.data_ptr(), C.byref(pp),
                      d_rn.data_ptr(), d_sm.data_ptr(), stream.cuda_stream))
    ms = timed(po)
    sms = [capi.PoseOptSummary.from_buffer_copy(r.tobytes()) for r in d_sm.cpu().numpy()]
    its = np.array([q.iterations for q in sms])
    blocks = np.array([q.n_residual_blocks for q in sms])
    flops = float(((its + 1) * blocks).sum()) * 330.0       # block evaluations x ~330 FP64 flops each
    tf = flops / (ms * 1e-3) / 1e12
    out.append({"key": "pose_opt", "workload": f"Optimizer::PoseOptimization: {F_} frames x {N_} features per call (Ceres trust-region LM restated, "
                            f"{its.mean():.1f} iterations on average), two frames per wavefront (four from 8192 frames on)",
                "value": F_ / (ms * 1e-3), "unit": "refinements/s",
                "roofline": {"bound": "fp64_vector", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "kernel": "pose_opt_rows_kernel", "kernel_ms_avg": ms,
                             "flops_per_launch": flops,
                             # HBM bytes per launch from the committed FETCH_SIZE / WRITE_SIZE passes of this launch size (case poseopt)
                             "traffic": pmc_traffic("pose_opt_", F_ * (N_ * 61 + 2 * 96 + 64)),
                             "algorithmic_bytes_per_launch": F_ * (N_ * 61 + 2 * 96 + 64),
                             "note": "~330 FP64 flops per residual-block evaluation x (iterations + 1) x blocks; the features of a frame "
                                     "(11 KB) stay in L1/L2: bound by the latency of its dependent FP64 chain, not by HBM"}})
    del d_b, d_w, d_l, d_u, d_T0, d_T, d_rn, d_sm

    # ---- FindMatchDirect for many current frames: warp prelude (affine, search level, 10x10 warp, 8x8 cut) + Align2D
    Wm, Hm, Lm, nfm, ncand = 640, 480, 5, 64, 800
    ws, hs, ss, offs, nb = capi.pyramid_layout(Wm, Hm, Lm)
    pitch = (nb + 255) // 256 * 256
    cur_pack, kf_pack, cols = np.zeros((8, pitch), np.uint8), np.zeros((8, pitch), np.uint8), []
    Tk8, Tc8 = np.zeros((8, 12)), np.zeros((8, 12))
    for i in range(8):
        scn = synth.make_scene(width=Wm, height=Hm, levels=Lm, n_patches=ncand, seed=900 + i, margin=30)
        for l in range(Lm):
            kf_pack[i, offs[l]:offs[l] + ws[l] * hs[l]] = scn.ref_pyr[l].reshape(-1)
            cur_pack[i, offs[l]:offs[l] + ws[l] * hs[l]] = scn.cur_pyr[l].reshape(-1)
        Tk8[i], Tc8[i] = scn.T_ref_w.reshape(12), scn.T_cur_w_true.reshape(12)
        Xc = scn.p_world @ scn.T_cur_w_true[:, :3].T + scn.T_cur_w_true[:, 3]
        pxc = np.stack([scn.cam.fx * Xc[:, 0] / Xc[:, 2] + scn.cam.cx, scn.cam.fy * Xc[:, 1] / Xc[:, 2] + scn.cam.cy], 1)
        cols.append((scn.px, scn.bearing, scn.p_world, pxc + np.random.default_rng(i).uniform(-1.0, 1.0, pxc.shape)))
    cam_m = capi.camera_struct(scn.cam)
    rep_ = nfm // 8
    Mm = nfm * ncand
    d_cur, d_kf = tdev(np.tile(cur_pack, (rep_, 1))), tdev(np.tile(kf_pack, (rep_, 1)))
    d_Tk, d_Tc = tdev(np.tile(Tk8, (rep_, 1))), tdev(np.tile(Tc8, (rep_, 1)))
    fr_idx = np.repeat(np.arange(nfm, dtype=np.int32), ncand)
    d_fr, d_kfi = tdev(fr_idx), tdev(fr_idx.copy())                       # candidate -> its current frame / its keyframe (one each)
    d_rp = tdev(np.tile(np.concatenate([c[0] for c in cols]), (rep_, 1)).astype(np.float32))
    d_rl = torch.zeros(Mm, dtype=torch.int32, device=dev)
    d_rb, d_pw = tdev(np.tile(np.concatenate([c[1] for c in cols]), (rep_, 1))), tdev(np.tile(np.concatenate([c[2] for c in cols]), (rep_, 1)))
    d_px0 = tdev(np.tile(np.concatenate([c[3] for c in cols]), (rep_, 1)))
    d_px = d_px0.clone()
    d_sl, d_cv = torch.zeros(Mm, dtype=torch.int32, device=dev), torch.zeros(Mm, dtype=torch.uint8, device=dev)
    d_scr = torch.empty(ctx.lib.dsdtm_match_candidates_scratch_bytes(Mm), dtype=torch.uint8, device=dev)
    wa, ha, sa, oa = (C.c_int * Lm)(*ws), (C.c_int * Lm)(*hs), (C.c_int * Lm)(*ss), (C.c_size_t * Lm)(*offs)

    def fmd():
        with torch.cuda.stream(stream):
            d_px.copy_(d_px0, non_blocking=True)
        ctx.check(ctx.lib.dsdtm_match_candidates_batch_device(
            ctx.handle, d_cur.data_ptr(), nfm, d_kf.data_ptr(), nfm, pitch, Lm, wa, ha, sa, oa, C.byref(cam_m), d_Tk.data_ptr(), d_Tc.data_ptr(),
            d_fr.data_ptr(), d_kfi.data_ptr(), d_rp.data_ptr(), d_rl.data_ptr(), d_rb.data_ptr(), d_pw.data_ptr(), Lm - 3, 10, Mm,
            d_scr.data_ptr(), d_px.data_ptr(), d_sl.data_ptr(), d_cv.data_ptr(), stream.cuda_stream))
    ms = timed(fmd)
    b_cand = 4 + 4 + 8 + 4 + 24 + 24 + 16 + 16 + 4 + 1                  # candidate columns in, pixel in/out, level + flag out (the warped patches stay in LDS)
    alg = Mm * b_cand + 2 * nfm * ws[0] * hs[0]                            # + level 0 of every keyframe and current frame once
    out.append({"key": "find_match_direct", "workload": f"Feature_Alignment::FindMatchDirect: {Mm} candidates of {nfm} current frames per call ({ncand} each, {Wm}x{Hm}): "
                            f"SolveAffineMatrix, GetBestSearchLevel, WarpAffine, GetPatchNoBoarder, Align2DGaussNewton (cap 10)",
                "value": Mm / (ms * 1e-3), "unit": "candidates/s", "us_per_frame": ms * 1e3 / nfm,
                "matched_fraction": float(d_cv.float().mean().item()),
                "roofline": roofline_block("match_kernel", alg, ms, None, Mm, b_cand, "candidate",
                                           {"note": "ONE launch (round 5; rounds 1-4: two, with 328 B of patches per candidate through HBM): lane = candidate "
                                                    "for the FP64 chain in the reference's operation order, thread = sample for the 10x10 patches into LDS, "
                                                    "four candidates per wavefront for Align2D; bound by dependent FP64 / float chains, the bytes are 105 "
                                                    "per candidate + the level-0 images once"})})
    del d_cur, d_kf, d_rp, d_rb, d_pw, d_px0, d_px, d_scr

    # ---- detector image work: 256 frames of 640x480x5 levels per call
    Wd, Hd, Ld, nfr = 640, 480, 5, 256
    ws, hs, ss, offs, nb = capi.pyramid_layout(Wd, Hd, Ld)
    pitch = (nb + 255) // 256 * 256
    packed = np.zeros((8, pitch), np.uint8)
    for i in range(8):
        pyr = synth.build_pyramid(np.clip(np.rint(synth.make_texture(Hd, Wd, 40 + i)), 0, 255).astype(np.uint8), Ld)
        for l in range(Ld):
            packed[i, offs[l]:offs[l] + ws[l] * hs[l]] = pyr[l].reshape(-1)
    cell = 25
    gc, gr = (Wd + cell - 1) // cell, (Hd + cell - 1) // cell
    G = gc * gr
    dp = capi.DetectParams(cell, gc, gr, Ld, 20, 5.0)
    d_pyr = torch.from_numpy(np.tile(packed, (nfr // 8, 1))).to(dev)
    d_score = torch.empty((nfr, pitch), dtype=torch.uint8, device=dev)
    d_key = torch.empty((nfr, G), dtype=torch.int64, device=dev)
    d_s = torch.empty((nfr, G), dtype=torch.float32, device=dev)
    d_x, d_y, d_lv = (torch.empty((nfr, G), dtype=torch.int32, device=dev) for _ in range(3))
    wa, ha, sa, oa = (C.c_int * Ld)(*ws), (C.c_int * Ld)(*hs), (C.c_int * Ld)(*ss), (C.c_size_t * Ld)(*offs)

    def det():
        ctx.check(ctx.lib.dsdtm_detect_cells_batch_device(ctx.handle, d_pyr.data_ptr(), pitch, nfr, Ld, wa, ha, sa, oa, None, C.byref(dp),
                                                          d_score.data_ptr(), d_key.data_ptr(), d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(),
                                                          d_lv.data_ptr(), stream.cuda_stream))
    ms = timed(det)
    b_fr = 3 * sum(ws[l] * hs[l] for l in range(Ld))     # pyramid read by the score pass, score map written, then read by the select pass
    out.append({"key": "detector", "workload": f"Feature_detector::detect, image part: {nfr} frames of {Wd}x{Hd}x{Ld} levels per call (FAST-10 score map, non-max, "
                            f"Shi-Tomasi, best corner per {cell}-px cell)",
                "value": nfr / (ms * 1e-3), "unit": "frames/s", "us_per_frame": ms * 1e3 / nfr,
                "cells_with_a_corner_per_frame": float((d_s > 5.0).sum().item()) / nfr,
                "roofline": roofline_block("fast_", nfr * b_fr, ms, None, nfr, b_fr, "frame",
                                           {"note": "three launches (score, select, decode); the score pass is bound by its min/max network "
                                                    "(~95 VALU instructions per pixel), the select pass by the Shi-Tomasi scoring of the survivors (four per round)"})})
    return out


def streamed_entry(torch, dev, ctx, cam, cam_struct, args, n_frames=2049, chunk=128):
    """The path fed from HOST memory (reference: src/Tracking.cpp:45-57 -> src/Frame.cpp:35-41,74-81 ->
    src/Sprase_ImageAlign.cpp:29-60) through ONE C-ABI call: dsdtm_sparse_align_batch_streamed on a chained sequence of
    n_frames level-0 images and their feature columns in pinned host memory. Inside the library: H2D in chunks on two copy
    streams -> pyramids on the device -> chained alignment (frame k is `cur` of pair k - 1 and `ref` of pair k, uploaded and
    built once) -> results D2H per chunk; upload of chunk j + 1 overlaps compute of chunk j. Reported: frames/s end to end
    (wall time of the call), the H2D rate against a bare hipMemcpyAsync of the same bytes on this box, the bytes against
    dsdtm_sparse_align_batch_sharded's full-pyramid upload of the same pairs, and that entry's time on the same pairs.
    PCIe-inclusive — never the headline value."""
    import torch.nn.functional as F
    from dsdtm_amd import capi, synth
    from tests import oracle_lib
    W, Hh, L, N = args.width, args.height, args.levels, args.patches
    ws, hs, strides, offs, pyr_bytes = capi.pyramid_layout(W, Hh, L, 64)
    pitch = (pyr_bytes + 255) // 256 * 256
    P = n_frames - 1
    rng = np.random.default_rng(0x5EC)
    # camera path in front of a textured plane z = depth (world = frame 0); frames rendered on the GPU
    depth = 2.0
    # a smooth closed path around the start (the plane stays in view): per-frame motion <= ~0.02 m / ~0.01 rad
    amp = np.array([0.15, 0.15, 0.08, 0.07, 0.07, 0.07])
    frq = rng.uniform(0.06, 0.13, 6)
    phs = rng.uniform(0, 2 * np.pi, 6)
    T = np.stack([synth.se3_exp(amp * np.sin(frq * k + phs)) for k in range(n_frames)])
    K = cam.K(); Kinv = np.linalg.inv(K); nrm = np.array([0.0, 0.0, 1.0])
    Hrc = np.stack([np.linalg.inv(K @ (T[k, :3, :3] + np.outer(T[k, :3, 3], nrm) / depth) @ Kinv) for k in range(n_frames)])
    tex = torch.from_numpy(synth.make_texture(Hh, W, 0x5EC).astype(np.float32)).to(dev)
    uu, vv = torch.meshgrid(torch.arange(W, device=dev, dtype=torch.float32), torch.arange(Hh, device=dev, dtype=torch.float32), indexing="xy")
    host_frames = torch.empty((n_frames, W * Hh), dtype=torch.uint8).pin_memory()
    for i0 in range(0, n_frames, 64):
        i1 = min(n_frames, i0 + 64)
        Hm = torch.tensor(Hrc[i0:i1], dtype=torch.float32, device=dev)
        den = Hm[:, 2, 0, None, None] * uu + Hm[:, 2, 1, None, None] * vv + Hm[:, 2, 2, None, None]
        xr = (Hm[:, 0, 0, None, None] * uu + Hm[:, 0, 1, None, None] * vv + Hm[:, 0, 2, None, None]) / den
        yr = (Hm[:, 1, 0, None, None] * uu + Hm[:, 1, 1, None, None] * vv + Hm[:, 1, 2, None, None]) / den
        grid = torch.stack([xr / (W - 1) * 2 - 1, yr / (Hh - 1) * 2 - 1], dim=-1)
        img = F.grid_sample(tex[None, None].expand(i1 - i0, -1, -1, -1), grid, mode="bicubic", padding_mode="reflection", align_corners=True)[:, 0]
        host_frames[i0:i1].copy_(img.round().clamp(0, 255).to(torch.uint8).reshape(i1 - i0, -1))
    del tex, uu, vv
    # features of every reference frame (host, pinned): pixels, bearings, points where the rays meet the plane
    px = np.stack([rng.uniform(30, W - 30, (P, N)), rng.uniform(30, Hh - 30, (P, N))], axis=2).astype(np.float32)
    bearing = synth.bearing_from_px(cam, px.reshape(-1, 2)).reshape(P, N, 3)
    R, t = T[:P, :3, :3], T[:P, :3, 3]
    Cw = -np.einsum("pji,pj->pi", R, t)
    dw = np.einsum("pji,pnj->pni", R, bearing)
    sd = (depth - Cw[:, None, 2]) / dw[:, :, 2]
    p_world = Cw[:, None, :] + dw * sd[:, :, None]
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    h = dict(px=pin(px), bearing=pin(bearing), p_world=pin(p_world), initial=pin(np.ones((P, N), np.uint8)),
             T_ref_w=pin(T[:P, :3, :].reshape(P, 12)))
    h_T = torch.empty((P, 12), dtype=torch.float64).pin_memory()
    h_nt = torch.zeros(P, dtype=torch.int32).pin_memory()
    h_st = torch.zeros((P, capi.STATS_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
    prm = capi.AlignParams(L, 0, args.iters, 15)
    sd_ = capi.StreamDesc()
    sd_.n_pairs, sd_.max_features, sd_.levels, sd_.width, sd_.height = P, N, L, W, Hh
    sd_.row_stride, sd_.image_pitch = W, W * Hh
    sd_.ref_image, sd_.cur_image = host_frames.data_ptr(), None              # chained: P + 1 frames
    sd_.px_xy, sd_.bearing, sd_.p_world, sd_.initial = (h[k].data_ptr() for k in ("px", "bearing", "p_world", "initial"))
    sd_.n_features, sd_.T_ref_w, sd_.T_cur_w = None, h["T_ref_w"].data_ptr(), h_T.data_ptr()
    sd_.n_tracked, sd_.stats = h_nt.data_ptr(), h_st.data_ptr()
    one = (C.c_void_p * 1)(ctx.handle)

    def run():
        h_T.copy_(h["T_ref_w"])                       # seed: cur.pose = ref.pose (src/Tracking.cpp:201); host-side, outside the clock
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_streamed(one, 1, C.byref(sd_), chunk, C.byref(cam_struct), C.byref(prm)))
        return time.perf_counter() - t0

    run()                                             # warm-up (allocations, clocks)
    t_pipe = min(run() for _ in range(3))
    T_pipe, nt_pipe = h_T.numpy().copy(), h_nt.numpy().copy()
    st_pipe = np.frombuffer(h_st.numpy().tobytes(), dtype=capi.STATS_DTYPE).copy()
    nbytes = host_frames.numel() + sum(v.numel() * v.element_size() for v in h.values()) + h_T.numel() * 8
    # the box's ceiling: one hipMemcpyAsync of the same number of bytes, pinned -> device, contiguous
    flat_h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    flat_d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    flat_d.copy_(flat_h, non_blocking=True); torch.cuda.synchronize()
    t_ceil = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        flat_d.copy_(flat_h, non_blocking=True); torch.cuda.synchronize()
        t_ceil = min(t_ceil, time.perf_counter() - t0)
    del flat_h, flat_d
    # the round-3 form of the same work: dsdtm_sparse_align_batch_sharded on host-built FULL pyramids of a 256-pair sample
    # (upload, one launch, download — nothing overlapped), scaled to the pair count for the byte comparison
    S = min(256, P)
    hostpyr = np.zeros((S + 1, pitch), np.uint8)
    d_tmp = torch.zeros((S + 1, pitch), dtype=torch.uint8, device=dev)
    d_tmp[:, :W * Hh].copy_(host_frames[:S + 1].to(dev))
    wa, ha, sa = (C.c_int * L)(*ws), (C.c_int * L)(*hs), (C.c_int * L)(*strides)
    oa = (C.c_size_t * L)(*offs)
    ctx.check(ctx.lib.dsdtm_pyrdown_batch_device(ctx.handle, d_tmp.data_ptr(), pitch, S + 1, L, wa, ha, sa, oa, None))
    torch.cuda.synchronize()
    hostpyr[...] = d_tmp.cpu().numpy()
    del d_tmp
    hp = torch.from_numpy(hostpyr).pin_memory()
    hb = capi.BatchDesc()
    hb.n_pairs, hb.max_features, hb.levels, hb.pyr_pitch = S, N, L, pitch
    for l in range(L):
        hb.width[l], hb.height[l], hb.stride[l], hb.level_offset[l] = ws[l], hs[l], strides[l], offs[l]
    To = h["T_ref_w"][:S].clone().pin_memory(); nto = torch.zeros(S, dtype=torch.int32).pin_memory()
    hb.ref_pyr, hb.cur_pyr = hp.data_ptr(), hp.data_ptr() + pitch
    hb.px_xy, hb.bearing, hb.p_world, hb.initial = (h[k].data_ptr() for k in ("px", "bearing", "p_world", "initial"))
    hb.n_features, hb.T_ref_w, hb.T_cur_w, hb.n_tracked, hb.stats = None, h["T_ref_w"].data_ptr(), To.data_ptr(), nto.data_ptr(), None
    t_sh = 1e9
    for _ in range(3):
        To.copy_(h["T_ref_w"][:S])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.check(ctx.lib.dsdtm_sparse_align_batch_sharded(one, 1, C.byref(hb), C.byref(cam_struct), C.byref(prm)))
        t_sh = min(t_sh, time.perf_counter() - t0)
    sharded_same = bool(np.array_equal(To.numpy(), T_pipe[:S]) and np.array_equal(nto.numpy(), nt_pipe[:S]))
    bytes_full = P * (2 * pitch + N * 57 + 2 * 96)
    # parity of a sample of the chained pairs against the CPU oracle on the same bytes
    lib = oracle_lib.load()
    sample = 16
    ob = capi.BatchDesc.from_buffer_copy(bytes(hb))
    ob.n_pairs = sample
    hx = {k: v[:sample].numpy().copy() for k, v in h.items()}
    Tor = hx["T_ref_w"].copy(); ntor = np.zeros(sample, np.int32); stor = np.zeros(sample, capi.STATS_DTYPE)
    ob.ref_pyr, ob.cur_pyr = hostpyr.ctypes.data, hostpyr.ctypes.data + pitch
    ob.px_xy, ob.bearing, ob.p_world, ob.initial = (hx[k].ctypes.data for k in ("px", "bearing", "p_world", "initial"))
    ob.n_features, ob.T_ref_w, ob.T_cur_w, ob.n_tracked, ob.stats = None, hx["T_ref_w"].ctypes.data, Tor.ctypes.data, ntor.ctypes.data, stor.ctypes.data
    lib.oracle_sparse_align_batch_timed(C.byref(ob), C.byref(cam_struct), C.byref(prm), usable_cpus())
    dl = np.array([synth.pose_error(T_pipe[i], Tor[i]) for i in range(sample)])
    err = np.array([synth.pose_error(T_pipe[i], T[i + 1, :3]) for i in range(P)])
    return {
        "key": "streamed_host_fed", "workload": f"streamed: {n_frames} chained {W}x{Hh} frames from pinned host memory ({P} pairs, {N} patches, {L} levels, cap "
                    f"{args.iters}) through ONE call of dsdtm_sparse_align_batch_streamed (C ABI): chunks of {chunk} pairs, H2D of level 0 + "
                    f"feature columns on 2 copy streams -> pyramids on the device -> chained alignment -> results D2H per chunk",
        "value": n_frames / t_pipe, "unit": "frames/s (PCIe-inclusive, end to end; = alignments/s + 1 frame)",
        "ms_total": t_pipe * 1e3,
        "h2d_bytes": int(nbytes), "h2d_achieved_GBps": nbytes / t_pipe / 1e9,
        "h2d_ceiling_GBps": nbytes / t_ceil / 1e9, "h2d_fraction_of_ceiling": t_ceil / t_pipe,
        "h2d_ceiling_note": "one hipMemcpyAsync of the same number of bytes, pinned host -> device, on this box; the call's wall time "
                            "also holds its allocation-free set-up, the last chunk's compute and the result download",
        "bytes_vs_full_pyramid_upload": {"streamed": int(nbytes), "sharded_full_pyramids": int(bytes_full),
                                         "fewer": 1.0 - nbytes / bytes_full,
                                         "note": "dsdtm_sparse_align_batch_sharded uploads both whole pyramids of every pair; the streamed entry "
                                                 "uploads level 0 of every FRAME once (chained) and builds the rest on the device"},
        "sharded_entry_same_pairs": {"pairs": S, "ms": t_sh * 1e3, "alignments_per_s": S / t_sh,
                                     "streamed_alignments_per_s": P / t_pipe, "results_bit_identical": sharded_same},
        "pose_delta_vs_cpu": {"max_rad": float(dl[:, 0].max()), "max_m": float(dl[:, 1].max()), "pairs_checked": sample,
                              "n_tracked_equal": bool(np.array_equal(nt_pipe[:sample], ntor)),
                              "iterations_equal": bool(np.array_equal(st_pipe["iters"][:sample], stor["iters"]))},
        "err_vs_ground_truth_median": {"rad": float(np.median(err[:, 0])), "m": float(np.median(err[:, 1]))}}


def fp64_block(args, kernel_ms):
    """The path's second roofline (SURVEY.md §8d asks for both): executed FP64 vector flops of the alignment kernel
    against the FP64 vector peak. Flops per launch come from the committed counter pass of the BASELINE workload and
    are null for any other shape."""
    flops = pmc_fp64_flops("sparse_align_reg_kernel") if (args.pairs, args.patches, args.width, args.height, args.levels, args.iters) == (1024, 300, 640, 480, 4, 10) else None
    out = {"bound": "fp64_vector", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "flops_per_launch": flops,
           "note": "executed FP64 vector flops per launch = 64 x (ADD + MUL + 2 FMA + TRANS) wave instructions, rocprofv3 "
                   "SQ_INSTS_VALU_*_F64 (" + PMC_SUMMARY + "); divided by this run's roofline.kernel_ms_avg = the kernel alone "
                   "(kernel_time_basis of the roofline block)"}
    if flops is not None:
        out["flops_per_alignment"] = flops / args.pairs
        out["achieved"] = flops / (kernel_ms * 1e-3) / 1e12
        out["frac"] = out["achieved"] / FP64_VECTOR_PEAK_TFLOPS
    else:
        out["achieved"] = out["frac"] = None
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N worker processes (one per GPU) BEFORE anything in
    this process touches the GPU, with the environment torch.distributed.run would give them. Rank 0's JSON line
    goes to our stdout; we exit with the worst worker status. (No exec: a child per rank, and we wait.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


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
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSDTM_BENCH_STREAMS", "8")),
                    help="HIP streams the steps are issued on in turn (default 8 = two per HIP hardware queue): the workgroups "
                         "of step k+1 take the compute units the tail of step k leaves idle (same launches, same "
                         "results); 1 = strictly one launch at a time, with HIP events around every launch")
    ap.add_argument("--preroll", type=int, default=300,
                    help="untimed launches of the same step BEFORE the W warm-up steps, reported in the line as 'preroll': the GPU "
                         "comes out of idle over ~40 ms of load (tools/warmup_sweep.sh: the same 20 timed steps run 7 %% faster "
                         "behind 200 launches than behind 5); 0 = none")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="pairs timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--rank-check-pairs", type=int, default=32,
                    help="N > 1: pairs of its OWN last step every rank checks against the CPU oracle after the timed region "
                         "(gathered into pose_delta_vs_cpu.ranks_checked; a rank over 1e-4 rad / m nulls value, exit 1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary roofline entries (other shapes, pyrDown)")
    ap.add_argument("--stub", action="store_true",
                    help="plumbing test without a GPU (tests/test_multiproc_cpu.py): gloo, no kernels, value null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import torch
    from dsdtm_amd import shard

    # DSDTM_BENCH_FORCE_DIST=1 (tests): the process group, its barrier and reductions also for ONE rank — the RCCL code path of
    # an N-GPU run exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("DSDTM_BENCH_FORCE_DIST") == "1"
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The data path has no collective; the process group only carries the contract's barrier and the
        # max-over-ranks reduction. RCCL ("nccl") when every rank has its own GPU; gloo for the CPU stub and for
        # DSDTM_BENCH_SHARE_GPU=1 (rehearsal of the multi-process path on a one-GPU box: all ranks use cuda:0,
        # RCCL refuses two ranks on one device).
        share_gpu = os.environ.get("DSDTM_BENCH_SHARE_GPU") == "1"
        if args.stub or share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if share_gpu:
            local_rank = 0

    def barrier():
        if use_dist:
            dist.barrier()

    if args.stub:
        # the launch path, the barrier and the max-over-ranks reduction without a GPU; never a measurement
        lo, hi = shard.pair_range(args.pairs * world, rank, world)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pass
        local = time.perf_counter() - t0
        barrier()
        elapsed = time.perf_counter() - t0
        per_rank = [[local, float(hi - lo)]]
        checks = [[0.0, 0.0, 1.0, 1.0, 0.0]]              # the shape of a rank_parity row; the stub checks no pair (pairs_checked 0)
        if use_dist:
            elapsed = shard.max_over_ranks(elapsed, dist, torch.device("cpu"))
            total = shard.sum_over_ranks(hi - lo, dist, torch.device("cpu"))
            per_rank = shard.gather_over_ranks([local, float(hi - lo)], dist, torch.device("cpu"))
            checks = shard.gather_over_ranks(checks[0], dist, torch.device("cpu"))
        else:
            total = hi - lo
        if rank == 0:
            # compact line only: the stub writes no bench_secondary.json (never a measurement)
            print(bench_line.compact_line({
                "metric": "stub", "value": None, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "data": "stub (no GPU work)", "pairs_per_step_all_ranks": total, "elapsed_max_s": elapsed,
                "per_rank_ms_per_step": [r[0] / max(1, args.steps) * 1e3 for r in per_rank],
                "per_rank_value": [None for _ in per_rank], "ranks_seen": len(per_rank),
                "pose_delta_vs_cpu": aggregate_rank_checks(checks)[0],
                **({"barrier_backend": dist.get_backend()} if use_dist else {})}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    from dsdtm_amd import capi, synth
    capi.diag_default(False).__enter__()          # every context of this process, explicit or implied: the release library
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if not os.path.exists(capi.lib_path()):   # tooling convenience only: the product itself never builds or falls back
        from dsdtm_amd.csrc import build as hip_build
        if rank == 0:
            hip_build.build(verbose=False)
        barrier()
    ctx = capi.Context(local_rank, diag=False)   # the RELEASE library whatever the environment says (DSDTM_PY_DIAG is for tools/);
                                                 # fails loudly without the HIP library / a gfx950 device
    cam = synth.Camera.tum(args.width, args.height)
    cam_struct = capi.camera_struct(cam)
    prm = capi.AlignParams(args.levels, 0, args.iters, 15)
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    stream = streams[0]

    d = build_batch(torch, dev, ctx, cam, args.pairs, args.width, args.height, args.levels, args.patches,
                    seed=shard.batch_seed(0xD5D7, rank), stream=stream)
    desc = d["desc"]

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
                                                          streams[k % n_streams].cuda_stream))

    per_kernel = n_streams == 1

    def timed_region():
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides.
        One stream: HIP events around every launch. Several streams: one pair around the whole timed region (its span is
        what a step costs the GPU; events around every launch were tried — they cost 4 % and, with other streams' kernels
        in flight, do not bracket the kernel). Returns (wall seconds, max over ranks; per-launch ms list or None; span ms)."""
        for k in range(args.warmup):
            step(k)
        for s in streams:
            s.synchronize()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps if per_kernel else 0)]
        span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        join = [torch.cuda.Event() for _ in streams]
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if not per_kernel:
            span[0].record(streams[0])            # span of the whole timed region on the launch streams
            for s in streams[1:]:
                s.wait_event(span[0])
        for k in range(args.steps):
            if per_kernel:
                ev[k][0].record(stream)           # HIP events on the stream the kernel is launched on
            step(args.warmup + k)
            if per_kernel:
                ev[k][1].record(stream)
        if not per_kernel:
            for i, s in enumerate(streams[1:], 1):
                join[i].record(s)
                streams[0].wait_event(join[i])
            span[1].record(streams[0])
        for s in streams:
            s.synchronize()
        torch.cuda.synchronize()
        el_local = time.perf_counter() - t0       # this rank's own K steps (before it waits for the others)
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            el = shard.max_over_ranks(el, dist, dev if dist.get_backend() == "nccl" else torch.device("cpu"))
        timed_region.local = el_local
        return el, ([x.elapsed_time(y) for x, y in ev] if per_kernel else None), (None if per_kernel else span[0].elapsed_time(span[1]))

    # For the record, the same W + K steps BEFORE the pre-roll, i.e. on a GPU that sat idle while the host prepared the
    # batch and through streams in their first use: 'value_from_idle' in the line (what --preroll 0 reports as the value)
    elapsed_idle = None
    if args.preroll > 0:
        elapsed_idle = timed_region()[0]
        d["T_steps"].copy_(d["T_seed"].unsqueeze(0).expand(n_slots, -1, -1))
        torch.cuda.synchronize()

    # Pre-roll: the same launches on pose buffers of their own, so that the W warm-up steps and the K timed steps run on
    # a GPU that is out of its idle clocks whatever W is (reported in the line; never part of the timed region)
    preroll_ms = 0.0
    if args.preroll > 0:
        T_pre = d["T_seed"].unsqueeze(0).repeat(n_streams, 1, 1).contiguous()
        pre = []
        for i in range(n_streams):
            dk = capi.BatchDesc.from_buffer_copy(bytes(desc))
            dk.T_cur_w = T_pre[i].data_ptr()
            pre.append(dk)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for k in range(args.preroll):
            i = k % n_streams
            with torch.cuda.stream(streams[i]):
                T_pre[i].copy_(d["T_seed"], non_blocking=True)
            ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(pre[i]), C.byref(cam_struct), C.byref(prm),
                                                              streams[i].cuda_stream))
        torch.cuda.synchronize()
        preroll_ms = (time.perf_counter() - tp) * 1e3

    elapsed, kernel_ms, span_ms = timed_region()
    # every rank's own figure, gathered AFTER the timed region: a straggler shows in the line (value itself stays the
    # whole job over the max-over-ranks time)
    per_rank = [[timed_region.local, float(args.pairs * args.steps)]]
    if use_dist:
        per_rank = shard.gather_over_ranks(per_rank[0], dist, dev if dist.get_backend() == "nccl" else torch.device("cpu"))
    ctx.check(ctx.lib.dsdtm_sparse_align_check(ctx.handle, stream.cuda_stream))   # no hand-over wait timed out in any launch
    d["T_cur_w"] = d["T_steps"][n_slots - 1].clone()  # the last step's results are the ones checked below
    if per_kernel:
        k_avg, k_min, k_extra = float(np.mean(kernel_ms)), float(np.min(kernel_ms)), {}
        k_basis = "HIP events around every launch on the launch stream (one stream: launches do not overlap)"
    else:
        k_avg = span_ms / args.steps
        k_min = None
        # a short one-stream burst OUTSIDE the timed region: the duration of the kernel when nothing overlaps it
        solo_n = 30
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(solo_n)]
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[0]):
            for k in range(solo_n):
                d["T_steps"][k % n_slots].copy_(d["T_seed"], non_blocking=True)      # re-seed (outside the event pair, same stream)
                evs[k][0].record(streams[0])
                ctx.check(ctx.lib.dsdtm_sparse_align_batch_device(ctx.handle, C.byref(descs[k % n_slots]), C.byref(cam_struct), C.byref(prm),
                                                                  streams[0].cuda_stream))
                evs[k][1].record(streams[0])
        streams[0].synchronize()
        k_solo = float(np.mean([a.elapsed_time(b) for a, b in evs][5:]))
        ov = pmc_summary().get("overlap", {})     # from the committed kernel trace of this binary (tools/profile.sh, case main)
        k_extra = {"span_ms_per_step": k_avg,
                   "kernel_ms_own_in_flight": ov.get("own_duration_avg_ns", 0) * 1e-6 or None,
                   "launches_in_flight": ov.get("launches_in_flight_avg"),
                   "in_flight_note": "a launch's own begin-to-end time and the number of launches in flight with the default streams: "
                                     "rocprofv3 --kernel-trace of this command, committed in " + PMC_SUMMARY + " (null when that "
                                     "summary was taken with another build of the library)",
                   "kernel_ms_solo": k_solo,
                   "kernel_ms_solo_note": f"mean of {solo_n - 5} launches on ONE stream after the timed region (HIP events around each): "
                                          "the kernel's duration as rocprofv3 --kernel-trace reports it for a one-stream run"}
        k_basis = (f"kernel_ms_avg = kernel_ms_solo: the kernel alone on one stream (HIP events around each launch of a burst right after "
                   f"the timed region) — the duration rocprofv3 --kernel-trace reports and profiles/ holds; frac is computed from it. The timed "
                   f"region issues its steps on {n_streams} streams (consecutive launches overlap at their edges): span_ms_per_step and "
                   f"frac_overlapped describe that delivered rate")
        k_extra["frac_overlapped"] = args.pairs * algorithmic_bytes(args.width, args.height, args.levels, args.patches) / (k_avg * 1e-3) / 1e9 / HBM_PEAK_GBS
        k_extra["frac_overlapped_note"] = ("algorithmic bytes / span per step of the timed region: what the GPU delivers with ~3 launches in flight; "
                                           "no single kernel's duration reproduces it (a launch's own begin-to-end time is kernel_ms_own_in_flight)")
        k_span = k_avg
        k_avg = k_solo

    # N > 1: EVERY rank holds its own last step against the CPU oracle (the checker runs after the timed region, on that rank's
    # share of the host threads), and the verdicts are gathered: an N-GPU rate is only a result if ranks 1..N-1 aligned their
    # pairs too. (N = 1: rank 0 checks all its pairs below, in the cpu_baseline leg.)
    rank_checks = None
    if world > 1 and args.rank_check_pairs > 0:
        n_chk = max(1, min(args.rank_check_pairs, args.pairs))
        if rank == 0:                         # the checker's library is (re)built by ONE rank if it is stale, never by all of them at once
            from tests import oracle_lib
            oracle_lib.load()
        barrier()
        mine = rank_parity(d, cam_struct, prm, n_chk, max(1, usable_cpus() // world))
        rank_checks = shard.gather_over_ranks(mine, dist, dev if dist.get_backend() == "nccl" else torch.device("cpu"))

    rc = 0
    if rank == 0:
        n_total = args.pairs * world * args.steps
        value = n_total / elapsed
        b_alg = algorithmic_bytes(args.width, args.height, args.levels, args.patches)
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
            "config": {"workload": f"{args.pairs} independent {args.width}x{args.height} pairs per GPU per step (BASELINE config 4 share of "
                                   f"config 2's shape), {args.levels} levels, {args.patches} patches, cap {args.iters}, one launch per step",
                       "pairs_per_gpu": args.pairs, "patches": args.patches, "levels": args.levels,
                       "max_iters": args.iters, "launch_streams": n_streams,
                       "parallelism": f"independent pairs x{world} (no collective)",
                       **({"barrier_backend": dist.get_backend()} if use_dist else {})},
            "per_rank_ms_per_step": [r[0] / args.steps * 1e3 for r in per_rank],
            "per_rank_value": [r[1] / r[0] for r in per_rank],
            "ranks_seen": dist.get_world_size() if use_dist else 1,
            **({"barrier_backend": dist.get_backend()} if use_dist else {}),
            "roofline": roofline_block("sparse_align_reg_kernel", args.pairs * b_alg, k_avg, k_min, args.pairs, b_alg, "alignment",
                                       {"kernel_time_basis": k_basis, **k_extra}),
            "fp64": fp64_block(args, k_avg),
            "executed_iterations_per_level_mean": [float(x) for x in iters.mean(axis=0)],
            "executed_iterations_total_mean": float(iters.sum(axis=1).mean()),
            "n_tracked_mean": float(ntg.mean()),
            "err_vs_ground_truth_median": {"rad": float(np.median(err[:, 0])), "m": float(np.median(err[:, 1]))},
            "library": ctx.lib.dsdtm_version().decode(), "library_sha": library_sha(),
            "value_from_idle": (n_total / elapsed_idle) if elapsed_idle else None,
            "preroll": {"launches": args.preroll, "ms": preroll_ms,
                        "note": "untimed launches of the same step before the 'warmup' steps, on pose buffers of their own: the GPU "
                                "leaves its idle clocks over ~40 ms of load (tools/warmup_sweep.sh, profiles/r03_warmup_sweep.txt); "
                                "value_from_idle = the same W + K steps timed before the pre-roll, in this process; "
                                "--preroll 0 makes that the value"},
        }
        if not args.no_cpu and world == 1:
            sample = min(args.cpu_sample, args.pairs)
            rows, To, nto, sto = cpu_baselines(d, cam_struct, prm, sample)
            out.update(rows)
            dl = np.array([synth.pose_error(Tg[i], To[i]) for i in range(sample)])
            pd = {"max_rad": float(dl[:, 0].max()), "max_m": float(dl[:, 1].max()), "pairs_checked": int(sample),
                  "n_tracked_equal": bool(np.array_equal(ntg[:sample], nto)),
                  "iterations_equal": bool(np.array_equal(stats["iters"][:sample], sto["iters"])),
                  "ranks_checked": 1, "tolerance": "1e-4 rad / 1e-4 m (north_star)"}
            out["pose_delta_vs_cpu"] = pd
            if not (pd["max_rad"] <= 1e-4 and pd["max_m"] <= 1e-4 and pd["n_tracked_equal"]):
                # a fast kernel with different results is not a result: no headline number, non-zero exit
                out["value"] = None
                out["parity_failed"] = True
                rc = 1
        if rank_checks is not None:
            pd, ok = aggregate_rank_checks(rank_checks)
            out["pose_delta_vs_cpu"] = pd
            if not ok:
                out["value"] = None
                out["parity_failed"] = True
                rc = 1
        if rc == 0 and world == 1 and not args.no_secondary:
            try:
                out["secondary"] = secondary_entries(torch, dev, ctx, cam_struct, stream, args)
            except Exception as e:          # the headline was measured and checked before this: it is printed whatever happens
                out["secondary_error"] = f"{type(e).__name__}: {e}"     # here — but a crashed secondary kernel is not a success:
                rc = 3                                                  # distinct non-zero exit (1 = parity of the headline failed)
        # secondary entries: one short line each, BEFORE the headline; full objects (with their notes) -> bench_secondary.json;
        # the LAST stdout line is the compact headline, < 4 KB (bench_line.py asserts it)
        bench_line.emit(out, ROOT)
    if use_dist:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
