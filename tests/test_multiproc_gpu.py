"""The multi-process GPU path of bench.py, rehearsed on ONE GPU: the code an 8-GPU SCALE run executes first.

`python bench.py --gpus 2` spawns one worker per rank BEFORE anything in the parent touches the GPU; the ranks
rendezvous on 127.0.0.1, build their own batches (seeded per rank), run the contract's barrier / synchronize /
max-over-ranks timing around real launches and rank 0 prints ONE line. With DSDTM_BENCH_SHARE_GPU=1 both ranks use
device 0 (gloo carries the barrier: RCCL refuses two ranks on one device), so the whole path except the RCCL backend
itself runs here. tests/test_multiproc_cpu.py covers the same plumbing without a GPU (--stub)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["DSDTM_BENCH_SHARE_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "64",
                        "--preroll", "4", "--no-cpu", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["pairs_per_gpu"] == 64 and out["config"]["barrier_backend"] == "gloo"
    # whole-job rate: both ranks' pairs over the max-over-ranks time
    assert out["value"] is not None and out["value"] > 0
    assert abs(out["value"] - 2 * 64 * 3 / (out["ms_per_step"] * 3e-3)) <= 2e-5 * out["value"]
    # rank 0's own results are sane (64 synthetic pairs converge to their ground truth)
    assert out["err_vs_ground_truth_median"]["rad"] < 1e-3 and out["n_tracked_mean"] > 250
    assert out["roofline"]["frac"] > 0
    # one figure per rank (gathered after the timed region): a straggler of an N-GPU run shows in the line
    assert out["ranks_seen"] == 2 and out["barrier_backend"] == "gloo"
    assert len(out["per_rank_ms_per_step"]) == len(out["per_rank_value"]) == 2
    assert all(0 < t <= out["ms_per_step"] * (1 + 1e-6) for t in out["per_rank_ms_per_step"])     # own time <= max-over-ranks time
    assert all(abs(v - 64 * 3 / (t * 3e-3)) <= 1e-4 * v for v, t in zip(out["per_rank_value"], out["per_rank_ms_per_step"]))
    assert len(lines[0]) < 4096
    # EVERY rank held pairs of its own last step against the CPU oracle after the timed region (the reference's result is
    # the pose, src/Sprase_ImageAlign.cpp:57-59): both verdicts are in the line
    pd = out["pose_delta_vs_cpu"]
    assert pd["ranks_checked"] == 2 and pd["pairs_checked"] == 64 and pd["max_rad"] <= 1e-4 and pd["max_m"] <= 1e-4
    assert pd["n_tracked_equal"] and pd["iterations_equal"]


@pytest.mark.gpu
def test_bench_one_rank_over_rccl():
    """The backend an N-GPU run uses — RCCL ("nccl") with one device per rank — exercised with ONE rank: process group with
    device_id, the contract's barrier on both sides of the timed region and the max-over-ranks all-reduce on the GPU
    (DSDTM_BENCH_FORCE_DIST=1 makes bench.py set them up for a world of 1; the two-rank test above has to use gloo because
    RCCL refuses two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "DSDTM_BENCH_SHARE_GPU")}
    env.update(DSDTM_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--pairs", "64",
                        "--preroll", "4", "--no-cpu", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["barrier_backend"] == "nccl" and out["barrier_backend"] == "nccl"
    assert out["ranks_seen"] == 1 and len(out["per_rank_ms_per_step"]) == 1          # gathered on the GPU over RCCL
    assert out["value"] > 0 and abs(out["value"] - 64 * 3 / (out["ms_per_step"] * 3e-3)) <= 2e-5 * out["value"]


@pytest.mark.gpu
def test_bench_two_ranks_under_the_drivers_launcher():
    """The same two ranks started the way the driver starts them (`python -m torch.distributed.run ... bench.py --gpus 2`), sharing
    device 0: environment from the launcher, one line from rank 0, whole-job rate over the max-over-ranks time."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["DSDTM_BENCH_SHARE_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "64",
                        "--preroll", "4", "--no-cpu", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["pairs_per_gpu"] == 64 and out["value"] > 0
    assert abs(out["value"] - 2 * 64 * 3 / (out["ms_per_step"] * 3e-3)) <= 2e-5 * out["value"]
    assert out["pose_delta_vs_cpu"]["ranks_checked"] == 2 and out["pose_delta_vs_cpu"]["max_rad"] <= 1e-4


@pytest.mark.gpu
def test_bench_last_line_is_compact_with_every_secondary_entry():
    """The driver's command with small sizes, secondary entries and the CPU legs included: the LAST stdout line is one JSON object
    under 4 KB with the contract's keys, `roofline` and `cpu_baseline`; every secondary entry is an earlier short line; the full
    objects are in bench_secondary.json (round 4's 20 KB line was not parsed by the driver)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "64",
                        "--cpu-sample", "8", "--preroll", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = r.stdout.splitlines()
    last = lines[-1]
    assert len(last) < 4096, len(last)
    out = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "cpu_baseline_all_cores", "pose_delta_vs_cpu", "value_from_idle", "preroll",
              "library", "per_rank_ms_per_step", "ranks_seen"):
        assert k in out, k
    assert out["steps"] == 2 and out["warmup"] == 1 and out["n_gpus"] == 1 and out["value"] > 0 and "secondary_error" not in out
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "hbm_frac_measured", "frac_overlapped"):
        assert k in out["roofline"], k
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0 and out["roofline"]["frac"] > 0
    assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert out["pose_delta_vs_cpu"]["max_rad"] <= 1e-4 and out["pose_delta_vs_cpu"]["max_m"] <= 1e-4 and out["pose_delta_vs_cpu"]["pairs_checked"] == 8
    sec = [json.loads(l) for l in lines[:-1] if l.startswith('{"secondary"')]
    assert len(sec) == out["secondary"]["entries"] >= 13 and all(len(l) < 1024 for l in lines[:-1] if l.startswith("{"))
    keys = {e["secondary"] for e in sec}
    for k in ("align_1024x1000_640x480", "align_256x2000_1280x960", "align_1024x190_640x480_L5_cap8", "pyramid", "align2d", "pose_opt", "find_match_direct", "detector",
              "run_one_pair_config2", "run_one_pair_config3", "run_one_pair_config5", "tracked_frame", "streamed_host_fed"):
        assert k in keys, (k, keys)
    with open(os.path.join(ROOT, "bench_secondary.json")) as f:
        full = json.load(f)
    assert len(full["secondary"]) == len(sec) and full["headline"]["roofline"]["kernel_time_basis"]
