"""N>1 plumbing of bench.py on CPU: two processes over gloo agree on the shard assignment and on
the max-over-ranks timing reduction (the data path itself has no collective: pairs are independent)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo_shards_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from dsdtm_amd import shard
        dist.init_process_group(backend="gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        lo, hi = shard.pair_range(8192, rank, world)
        seed = shard.batch_seed(0xD5D7, rank)
        elapsed = shard.max_over_ranks(0.25 * (rank + 1), dist, torch.device("cpu"))
        total = shard.sum_over_ranks(hi - lo, dist, torch.device("cpu"))
        dist.barrier()
        print(json.dumps(dict(rank=rank, lo=lo, hi=hi, seed=seed, elapsed=elapsed, total=total)), flush=True)
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        outs.append(eval(o.strip().splitlines()[-1].replace("true", "True")))
    outs.sort(key=lambda d: d["rank"])
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 4096, 4096, 8192)
    assert outs[0]["seed"] != outs[1]["seed"]
    assert outs[0]["elapsed"] == outs[1]["elapsed"] == 0.5          # max over ranks
    assert outs[0]["total"] == outs[1]["total"] == 8192


def test_pair_range_covers_everything_once():
    sys.path.insert(0, ROOT)
    from dsdtm_amd import shard
    for n, w in [(8192, 8), (1000, 3), (5, 8), (0, 4)]:
        got = []
        for r in range(w):
            lo, hi = shard.pair_range(n, r, w)
            got += list(range(lo, hi))
        assert got == list(range(n))


def test_bench_gpus_flag_spawns_one_worker_per_rank():
    """`python bench.py --gpus 2` without a launcher starts two ranks (spawned before anything touches a GPU) that
    rendezvous, run the barrier / max-over-ranks path and print ONE line with n_gpus = 2. --stub replaces the
    GPU steps (this box has no GPU); the spawn path, the environment and the reductions are the real ones."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "4", "--warmup", "1",
                        "--pairs", "1024"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["pairs_per_step_all_ranks"] == 2048
    assert out["value"] is None and "stub" in out["data"]          # never mistaken for a measurement
    # one entry per rank, gathered after the timed region (all_gather over the process group)
    assert out["ranks_seen"] == 2 and len(out["per_rank_ms_per_step"]) == 2 and out["barrier_backend"] == "gloo"
    assert len(lines[0]) < 4096
    # the per-rank parity rows travel the same way (all_gather, folded by rank 0); the stub checks no pair and says so
    assert out["pose_delta_vs_cpu"]["ranks_checked"] == 0 and out["pose_delta_vs_cpu"]["pairs_checked"] == 0


def test_bench_refuses_a_world_that_contradicts_the_flag():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_under_the_drivers_launcher():
    """The command line the driver uses for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` — with the stub steps: the ranks take RANK / LOCAL_RANK / WORLD_SIZE from the
    launcher's environment (no second spawn), and rank 0 alone prints the line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "4",
                        "--warmup", "1", "--pairs", "1024"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["pairs_per_step_all_ranks"] == 2048 and out["value"] is None
    assert out["ranks_seen"] == 2 and len(out["per_rank_ms_per_step"]) == 2
