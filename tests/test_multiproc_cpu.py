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
