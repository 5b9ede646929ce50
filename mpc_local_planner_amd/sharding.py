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


def run_sharded_job(leg, steps: int, warmup: int, rank: int, world: int, total: int, device=None, solo_reference: bool = True, force_gather: bool = False):
    """The N > 1 protocol of bench.py, as ONE function (bench.py calls it on the GPU legs over RCCL; tests/test_multi_gpu_cpu.py drives the same code under gloo with a CPU
    stand-in for the leg): `total` independent planner instances are partitioned over `world` ranks (shard_range: contiguous, sizes differ by at most one), every rank owns
    `leg` = its shard's solver + resident inputs / outputs.  No collective touches the data path.
      1. `warmup` untimed steps; 2. (solo_reference) rank 0 alone runs `steps` steps while the others wait at a barrier: the per-GPU denominator for scaling efficiency,
      measured in the same process state; 3. barrier, EXACTLY `steps` timed steps (each ends when its results are available: leg.step_wait()), barrier; the job's time is the
      MAXIMUM over the ranks; 4. outside the timed region, every rank receives the whole job's status / dt / x by all_gather (ragged shards padded) and checks that its own
      slice of the gathered arrays is what it computed.
    leg: .step_wait() (one step, returns when its results are available), .sync() (everything queued is done), .results() -> (status, dt, x) arrays or tensors of this rank's
    shard.  Returns a dict: elapsed (max over ranks, seconds), solo_elapsed (rank 0, or None), converged_total, shard (lo, hi), gather_ms, gathered (status, dt, x)."""
    import time
    import torch.distributed as dist
    multi = dist.is_initialized() and (world > 1 or force_gather)
    lo, hi = shard_range(total, world, rank)
    for _ in range(warmup):
        leg.step_wait()
    leg.sync()
    solo_elapsed = None
    if multi and solo_reference:
        dist.barrier()
        leg.sync()
        if rank == 0:
            ts = time.perf_counter()
            for _ in range(steps):
                leg.step_wait()
            leg.sync()
            solo_elapsed = time.perf_counter() - ts
        dist.barrier()
    leg.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        leg.step_wait()
    leg.sync()
    if multi:
        dist.barrier()
    leg.sync()
    elapsed = max_over_ranks(time.perf_counter() - t0, device=device)
    st, dt, x = leg.results()
    assert st.shape[0] == hi - lo, (st.shape, lo, hi)
    out = {"elapsed": elapsed, "solo_elapsed": solo_elapsed, "shard": (lo, hi), "gather_ms": None, "gathered": None}
    is_np = isinstance(st, np.ndarray)
    n_conv = int((st == 0).sum()) if is_np else int((st == 0).sum().item())
    if multi:
        tg = time.perf_counter()
        g_st = gather_results(st, world, total, device=device, force=force_gather)
        g_dt = gather_results(dt, world, total, device=device, force=force_gather)
        g_x = gather_results(x, world, total, device=device, force=force_gather)
        if not is_np and g_x.is_cuda:
            import torch
            torch.cuda.synchronize()
        out["gather_ms"] = (time.perf_counter() - tg) * 1e3
        eq = (lambda a, b: np.array_equal(a, b)) if is_np else (lambda a, b: bool((a == b).all().item()))
        assert g_st.shape[0] == total and g_dt.shape[0] == total and g_x.shape[0] == total
        assert eq(g_st[lo:hi], st) and eq(g_x[lo:hi], x) and eq(g_dt[lo:hi], dt), "a rank's own slice of the gathered results differs from what it computed"
        n_conv = int((g_st == 0).sum()) if is_np else int((g_st == 0).sum().item())
        out["gathered"] = (g_st, g_dt, g_x)
    out["converged_total"] = n_conv
    return out
