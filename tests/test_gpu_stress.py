"""-m gpu: (i) determinism stress test of the candidate hand-off at BASELINE configs[3]'s whole batch on one GPU (during an A/B
run of round 2 one of ~40 launches at B = 32768 returned a different converged candidate for one instance; the hand-off of a hedge's record is now
release/acquire by construction, mpc_solve_kernel.hpp, and this test is the guard); (ii) the N > 1 code path of bench.py on every GPU box: process group on
RCCL with a world of one rank, all-gather of the device-resident results."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("these tests need the MI355X (no HIP device here)")
    torch.zeros(1, device="cuda")
    import mpc_local_planner_amd as pkg
    return pkg


def test_candidate_results_are_bit_identical_over_200_launches_at_B32768(m):
    """B = 32768 instances x 4 candidates = 131072 workgroups per launch (hedges start whenever a SIMD frees up: every launch has its own timing),
    200 launches, device resident: x / u / dt / status / iterations / winner of every launch are compared BIT FOR BIT with launch 0 on the device;
    any difference is counted per launch and per array.  ~70 ms per launch.  (SLOW_TIER)"""
    import torch
    n, B, LAUNCHES = 50, 32768, 200
    dev = torch.device("cuda", 0)
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
    s = m.BatchSolver(m.config_carlike_min_time(n, candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)), max_batch=B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_in = [t(x0), t(xf), t(up), t(dtp)]
    mk = lambda: dict(x=torch.full((B, n, 3), float("nan"), dtype=torch.float64, device=dev), u=torch.full((B, n, 2), float("nan"), dtype=torch.float64, device=dev),
                      dt=torch.full((B,), float("nan"), dtype=torch.float64, device=dev), st=torch.full((B,), -7, dtype=torch.int32, device=dev),
                      it=torch.full((B,), -7, dtype=torch.int32, device=dev))

    def launch(o):
        torch.cuda.current_stream().synchronize()      # the fills of `o` run on torch's stream, the solve on the handle's own (non-blocking) one: without this a late fill can overwrite what the solve wrote
        s.solve_device(B, d_in[0].data_ptr(), d_in[1].data_ptr(), d_in[2].data_ptr(), d_in[3].data_ptr(), None, None, None,
                       o["x"].data_ptr(), o["u"].data_ptr(), o["dt"].data_ptr(), o["st"].data_ptr(), o["it"].data_ptr())
        s.synchronize()
        w, tot = s.last_candidates(B)
        return w
    ref = mk()
    w_ref = launch(ref)
    assert (ref["st"] == 0).float().mean().item() > 0.99 and not torch.isnan(ref["x"]).any()
    bad = []
    for k in range(1, LAUNCHES):
        o = mk()              # fresh, poisoned output arrays: a result that is not written shows up
        w = launch(o)
        diffs = {name: int((o[name] != ref[name]).sum().item()) for name in ("x", "u", "dt", "st", "it")}
        diffs["winner"] = int((w != w_ref).sum())
        if any(diffs.values()):
            inst = np.nonzero(w != w_ref)[0][:4].tolist() or torch.nonzero((o["x"] != ref["x"]).reshape(B, -1).any(1))[:4, 0].tolist()
            bad.append((k, diffs, inst))
    print(f"B = {B}, {LAUNCHES} launches, winners histogram {np.bincount(w_ref[w_ref >= 0], minlength=4).tolist()} (+ {int((w_ref < 0).sum())} without a winner): "
          f"{len(bad)} launches differ from launch 0" + (f": {bad[:5]}" if bad else ""))
    assert not bad, bad[:5]
    s.close()


def test_global_form_blocks_are_claimed_and_released_without_a_trace_over_40_launches(m):
    """The global form of the factorisation data (mpc_config.stage_data) hands every workgroup a block of global memory claimed from the pool of ITS XCD
    (mpc_solve_kernel.hpp: HW_REG_XCC_ID + a CAS probe) -- 2048 blocks for 16384 workgroups per launch here, so every block changes hands about eight times per launch, in an
    order that differs from launch to launch.  A block that moved between XCDs, a claim that two workgroups won, or a stale word that some path consumed would show up as
    a difference: 40 launches of the config-5 shape (n = 120, fp64, four candidates, B = 4096) and 10 of config 3 (n = 80, 16 polygons, B = 4096) are compared bit for
    bit with launch 0 on the device, and launch 0 with the LDS form.  (SLOW_TIER)"""
    import torch
    from mpc_local_planner_amd import _abi as A
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    C5 = dict(candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0))
    B = 4096
    x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(B, n_obst=16, max_vertices=6, lateral=(0.15, 0.8))
    cases = [("config 5 shape", 120, 40, lambda **k: m.config_bicycle_min_time(120, **C5, **k), m.workloads.bicycle_min_time_inputs(B), None),
             ("config 3", 80, 10, lambda **k: m.config_unicycle_quadratic(80, max_obstacles=16, max_vertices=6, max_obstacle_rows=4, max_iter=60, **k), (x0, xf, up, dtp), obs)]
    for tag, n, launches, mk, inp, ob in cases:
        d_in = [t(a) for a in inp]
        d_ob = None if ob is None else [t(a) for a in ob]
        obp = None if ob is None else tuple(a.data_ptr() for a in d_ob)
        out = lambda: dict(x=torch.full((B, n, 3), float("nan"), dtype=torch.float64, device=dev), u=torch.full((B, n, 2), float("nan"), dtype=torch.float64, device=dev),
                           dt=torch.full((B,), float("nan"), dtype=torch.float64, device=dev), st=torch.full((B,), -7, dtype=torch.int32, device=dev),
                           it=torch.full((B,), -7, dtype=torch.int32, device=dev))

        def launch(s, o):
            torch.cuda.current_stream().synchronize()      # the fills of `o` run on torch's stream, the solve on the handle's own (non-blocking) one: without this a late fill can overwrite what the solve wrote
            s.solve_device(B, d_in[0].data_ptr(), d_in[1].data_ptr(), d_in[2].data_ptr(), d_in[3].data_ptr(), None, None, None,
                           o["x"].data_ptr(), o["u"].data_ptr(), o["dt"].data_ptr(), o["st"].data_ptr(), o["it"].data_ptr(), obstacles=obp)
            s.synchronize()
        sl = m.BatchSolver(mk(stage_data=A.STAGE_LDS), max_batch=B)
        lds = out(); launch(sl, lds); sl.close()
        s = m.BatchSolver(mk(), max_batch=B)                      # MPC_STAGE_AUTO: the global form on both workloads
        assert s.lds_bytes() < 45000
        ref = out(); launch(s, ref)
        assert (ref["st"] == 0).float().mean().item() > 0.95 and not torch.isnan(ref["x"]).any()
        assert all(torch.equal(ref[k], lds[k]) for k in ref), tag
        bad = []
        for k in range(1, launches):
            o = out(); launch(s, o)
            diffs = {name: int((o[name] != ref[name]).sum().item()) for name in ref}
            if any(diffs.values()):
                bad.append((k, diffs))
        print(f"[global-form blocks, {tag}] B = {B}, {launches} launches: {len(bad)} differ from launch 0" + (f": {bad[:5]}" if bad else ""))
        assert not bad, bad[:5]
        s.close()


def test_rccl_all_gather_of_device_resident_results_with_a_world_of_one_rank(m):
    """bench.py's N > 1 path on a one-GPU box: `nccl` (= RCCL) process group, a solve of the shard, all_gather of status / dt / x where they live
    (HBM to HBM), the own shard found in the gathered arrays."""
    import torch
    import torch.distributed as dist
    from mpc_local_planner_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dev = torch.device("cuda", 0)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
    try:
        n, B = 50, 512
        x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=sharding.rank_seed(20260924, 0))
        s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d = [t(x0), t(xf), t(up), t(dtp)]
        xo = torch.zeros((B, n, 3), dtype=torch.float64, device=dev); uo = torch.zeros((B, n, 2), dtype=torch.float64, device=dev)
        do = torch.zeros(B, dtype=torch.float64, device=dev); st = torch.full((B,), -1, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
        dist.barrier()
        s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize()
        torch.cuda.synchronize()
        g_st = sharding.gather_results(st, 1, B, force=True)
        g_dt = sharding.gather_results(do, 1, B, force=True)
        g_x = sharding.gather_results(xo, 1, B, force=True)
        torch.cuda.synchronize()
        lo, hi = sharding.shard_range(B, 1, 0)
        assert g_st.is_cuda and g_x.is_cuda and torch.equal(g_st[lo:hi], st) and torch.equal(g_x[lo:hi], xo) and torch.equal(g_dt[lo:hi], do)
        assert (g_st == 0).float().mean().item() > 0.9
        assert sharding.max_over_ranks(1.25, device=dev) == 1.25
        s.close()
    finally:
        if own:
            dist.destroy_process_group()
