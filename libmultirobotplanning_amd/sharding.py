"""Instance sharding across ranks (one process per GPU).

The path shards by instance — a low-level search, and a conflict tree, never spans ranks — so there is no data-path
collective: rank r generates/solves its own instances and only totals are reduced at the end.
"""
from typing import List, Sequence, Tuple


def seed_base(agents: int, rank: int, steps_per_rank: int, step_idx: int, batch: int) -> int:
    """First seed of the batch a rank solves at a step: disjoint across (rank, step); 1000*agents + k."""
    return 1000 * agents + (rank * steps_per_rank + step_idx) * batch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Strong-scaling split of a fixed instance list: item k goes to rank k % world."""
    return list(range(rank, n_items, world))


def reduce_totals(dist, device, elapsed: float, sums: Sequence[float]) -> Tuple[float, List[float]]:
    """max over ranks of the elapsed time, sum over ranks of the counters (RCCL on GPUs, gloo on CPU)."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    s = torch.tensor(list(sums), dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in s.tolist()]
