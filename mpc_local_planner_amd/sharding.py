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


def gather_results(local, world: int, total: int, device=None, force: bool = False):
    """all_gather of a per-rank result array along axis 0 into the global array (ragged shards allowed).
    `local` is a numpy array or a torch tensor.  A device-resident tensor is gathered where it lives (RCCL moves HBM to HBM); a numpy
    array / CPU tensor is staged on `device` first when the process group's backend is nccl (= RCCL, which only moves device memory).
    Returns the same kind of object it was given."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:      # force: run the collective even in a world of one rank (developer check of the RCCL path)
        return local
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local
    if not t.is_cuda and dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    sizes = [shard_range(total, world, r) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxlen,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    full = torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)
    return full.cpu().numpy() if is_np else full


def max_over_ranks(value: float, device=None) -> float:
    """Timing protocol of bench.py: the job takes as long as its slowest rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
