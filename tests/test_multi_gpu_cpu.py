"""N>1 host path on CPU: two gloo ranks partition a batch, each solves its shard (with the C oracle standing in
for the GPU, which does not exist here), results are all-gathered and must equal the single-process solve; the
timing reduction takes the maximum over ranks."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mpc_local_planner_amd import sharding, workloads
    from oracle import c_oracle as CO, se2_nlp as R
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(total, world, rank)
    x0, xf, up, dtp = workloads.carlike_min_time_inputs(total, seed=31, goal_range=(1.0, 2.5))
    oc = CO.from_nlp_config(R.config_carlike_min_time(12))
    xo, uo, do, st, it = CO.solve_batch(oc, x0[lo:hi], xf[lo:hi], up[lo:hi], dtp[lo:hi], nthreads=1)
    gx = sharding.gather_results(xo, world, total)
    gs = sharding.gather_results(st, world, total)
    tmax = sharding.max_over_ranks(1.0 + rank)
    if rank == 0:
        np.savez(tmp, x=gx, st=gs, tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path, c_oracle):
    from mpc_local_planner_amd import sharding, workloads
    from oracle import se2_nlp as R
    total, world = 11, 2          # ragged: 6 + 5
    assert [sharding.shard_range(total, world, r) for r in range(world)] == [(0, 6), (6, 11)]
    assert sharding.shard_range(7, 3, 2) == (5, 7)
    assert sharding.rank_seed(100, 3) == 103
    out = str(tmp_path / "res.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, total, out), nprocs=world, join=True)
    g = np.load(out)
    x0, xf, up, dtp = workloads.carlike_min_time_inputs(total, seed=31, goal_range=(1.0, 2.5))
    oc = c_oracle.from_nlp_config(R.config_carlike_min_time(12))
    xo, uo, do, st, it = c_oracle.solve_batch(oc, x0, xf, up, dtp, nthreads=1)
    np.testing.assert_array_equal(g["x"], xo)
    np.testing.assert_array_equal(g["st"], st)
    assert float(g["tmax"]) == 2.0
