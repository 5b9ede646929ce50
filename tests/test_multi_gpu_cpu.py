"""N>1 host path on CPU: gloo ranks partition a batch, each solves its shard (with the C oracle standing in
for the GPU, which does not exist here), results are all-gathered and must equal the single-process solve; the
timing reduction takes the maximum over ranks.  test_bench_protocol_at_world_8 drives sharding.run_sharded_job --
the function bench.py --gpus N runs on the GPU legs -- at world 8 on a ragged batch."""
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


class _OracleLeg:
    """CPU stand-in for bench.py's Leg: this rank's shard solved by the C oracle, inputs drawn from seed + rank (the weak-scaling rule of bench.py)"""

    def __init__(self, rank, n_inst, slow=0.0):
        from mpc_local_planner_amd import sharding, workloads
        from oracle import c_oracle as CO, se2_nlp as R
        self.CO, self.oc = CO, CO.from_nlp_config(R.config_carlike_min_time(12))
        self.inp = workloads.carlike_min_time_inputs(n_inst, seed=sharding.rank_seed(31, rank), goal_range=(1.0, 2.5))
        self.slow, self.steps_done, self.out = slow, 0, None

    def step_wait(self):
        import time
        self.out = self.CO.solve_batch(self.oc, *self.inp, nthreads=1)
        time.sleep(self.slow)
        self.steps_done += 1

    def sync(self):
        pass

    def results(self):
        xo, uo, do, st, it = self.out
        return st, do, xo


def _bench_worker(rank, world, port, total, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mpc_local_planner_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(total, world, rank)
    leg = _OracleLeg(rank, hi - lo, slow=0.02 * rank)          # the higher the rank, the slower: the job's time must be the slowest rank's
    res = sharding.run_sharded_job(leg, steps=3, warmup=1, rank=rank, world=world, total=total)
    assert leg.steps_done == 1 + 3 + (3 if rank == 0 else 0)        # warm-up + timed steps (+ rank 0's solo reference)
    assert res["shard"] == (lo, hi) and res["gather_ms"] is not None
    g_st, g_dt, g_x = res["gathered"]
    np.savez(tmp + f".{rank}", st=g_st, dt=g_dt, x=g_x, elapsed=res["elapsed"], solo=-1.0 if res["solo_elapsed"] is None else res["solo_elapsed"], conv=res["converged_total"])
    dist.barrier()
    dist.destroy_process_group()


def test_bench_protocol_at_world_8(tmp_path, c_oracle):
    """sharding.run_sharded_job (what bench.py --gpus N executes over RCCL) under gloo at world 8 on 43 instances (ragged: 6 6 6 5 5 5 5 5), inputs from seed + rank: every
    rank ends with the whole job's results, equal to each rank's own solve; the timing is the maximum over the ranks; only rank 0 measures the solo reference"""
    from mpc_local_planner_amd import sharding, workloads
    from oracle import se2_nlp as R
    total, world = 43, 8
    out = str(tmp_path / "res")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_bench_worker, args=(world, port, total, out), nprocs=world, join=True)
    oc = c_oracle.from_nlp_config(R.config_carlike_min_time(12))
    want_st, want_dt, want_x = [], [], []
    for r in range(world):
        lo, hi = sharding.shard_range(total, world, r)
        xo, uo, do, st, it = c_oracle.solve_batch(oc, *workloads.carlike_min_time_inputs(hi - lo, seed=sharding.rank_seed(31, r), goal_range=(1.0, 2.5)), nthreads=1)
        want_st.append(st); want_dt.append(do); want_x.append(xo)
    want_st, want_dt, want_x = np.concatenate(want_st), np.concatenate(want_dt), np.concatenate(want_x)
    el = []
    for r in range(world):
        g = np.load(out + f".{r}.npz")
        np.testing.assert_array_equal(g["st"], want_st); np.testing.assert_array_equal(g["dt"], want_dt); np.testing.assert_array_equal(g["x"], want_x)
        assert int(g["conv"]) == int((want_st == 0).sum())
        assert (float(g["solo"]) > 0) == (r == 0)
        el.append(float(g["elapsed"]))
    assert max(el) == min(el) and el[0] >= 3 * 0.02 * (world - 1)          # every rank reports the same job time, and it is the slowest rank's
