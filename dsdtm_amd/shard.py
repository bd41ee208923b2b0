"""Multi-GPU sharding of independent frame pairs: contiguous blocks of
ceil(P/G) pairs per rank, no data-path collective. The only communication is the benchmark's
barrier and the max-over-ranks of the elapsed time."""
from __future__ import annotations


def pair_range(n_pairs: int, rank: int, world: int) -> tuple[int, int]:
    per = -(-n_pairs // world) if world > 0 else n_pairs
    lo = min(n_pairs, rank * per)
    return lo, min(n_pairs, lo + per)


def batch_seed(base: int, rank: int) -> int:
    return base + 1000 * rank


def max_over_ranks(value: float, dist, device) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: int, dist, device) -> int:
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_over_ranks(values, dist, device) -> list:
    """Every rank's list of floats, in rank order, on every rank (bench.py: per-rank elapsed time and pair count)."""
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.cpu()] for o in out]
