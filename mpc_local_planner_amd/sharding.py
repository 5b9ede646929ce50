"""Multi-GPU host logic: independent planner instances are partitioned over ranks (one process per GPU);
there is no collective on the data path.  torch.distributed is used only to (optionally)
collect results and to reduce the timing."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` instances owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """Weak-scaling inputs: rank r draws its own instances from seed + r (no scatter needed)."""
    return seed + rank


def gather_results(local: np.ndarray, world: int, total: int):
    """all_gather of a per-rank result array along axis 0 into the global array (ragged shards allowed)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [shard_range(total, world, r) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = np.zeros((maxlen,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    t = torch.from_numpy(pad)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.numpy()[: hi - lo] for o, (lo, hi) in zip(out, sizes)], axis=0)


def max_over_ranks(value: float, device=None) -> float:
    """Timing protocol of bench.py: the job takes as long as its slowest rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
