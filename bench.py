#!/usr/bin/env python
"""bench.py -- MPC control-cycle solves/sec of the car-like minimum-time NLP (n=50) on MI355X.

One "step" = one pass of the hot path (a full batched NLP solve, cold start exactly as
Controller::step does on an empty grid) over one batch of synthetic planner inputs that are
already resident in HBM.  N=1 workload: BASELINE.json configs[1] (batch 1024 per GPU).  N>1:
independent planner instances are sharded over the ranks (weak scaling, 1024 per GPU, rank r draws
its inputs from seed+r); there is no data-path collective -- RCCL is used only for the barrier and
the max-over-ranks reduction of the timing.

Contract: python bench.py --gpus N --steps K --warmup W   (torchrun for N>1) -> ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_GRID = 50
BATCH_PER_GPU = 1024
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6     # SURVEY.md 8d / AMD spec, vector fp64


def algorithmic_bytes_per_solve(n: int) -> int:
    """B_alg = 8 * (2*P(n) + 9), P(n) = 5n - 1 scalars per trajectory (read guess + write solution)
    + x0(3) + xf(3) + u_prev(2) + dt_prev(1)."""
    return 8 * (2 * (5 * n - 1) + 9)


def cpu_baseline(n):
    """Times the C oracle (oracle/mpc_oracle.c, banded-LU interior point, OpenMP over instances) on a bounded
    sample of the same workload on the host cores.  Checker/baseline only: nothing here feeds the GPU path.
    The thread count is the one that gives the highest throughput in a short pilot (containers often expose more logical
    CPUs than their CPU quota lets them run; oversubscribing costs the oracle up to 2x)."""
    from oracle import c_oracle as CO, se2_nlp as R
    import mpc_local_planner_amd.workloads as W
    CO.build()
    oc = CO.from_nlp_config(R.config_carlike_min_time(n))
    logical = CO.num_threads()
    pilot = 2048
    x0, xf, up, dtp = W.carlike_min_time_inputs(pilot)
    CO.solve_batch(oc, x0[:64], xf[:64], up[:64], dtp[:64])      # untimed: library load + OpenMP team start-up
    best, cores, nt = 0.0, logical, logical
    while nt >= 4:
        t = time.perf_counter()
        CO.solve_batch(oc, x0, xf, up, dtp, nthreads=nt)
        rate = pilot / (time.perf_counter() - t)
        if rate > best:
            best, cores = rate, nt
        nt //= 2
    # ~5 s wall at the best thread count: with the long-tailed iteration counts (p50 28, max 100) a thread needs a few hundred
    # instances before its throughput stops depending on which instances it drew; more than one GPU batch = further draws
    # from the same distribution
    sample2 = int(min(64 * BATCH_PER_GPU, max(pilot, best * 5.0)))
    x0, xf, up, dtp = W.carlike_min_time_inputs(sample2)
    t = time.perf_counter()
    out = CO.solve_batch(oc, x0, xf, up, dtp, nthreads=cores)
    dt = time.perf_counter() - t
    return {"value": sample2 / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{sample2} instances drawn from the config-2 distribution (seed {W.SEED_CONFIG2}), cold start, tol 1e-8, "
                      f"mean {float(out[4].mean()):.1f} iterations, {dt:.2f} s wall on {cores} OpenMP threads "
                      f"(best of a pilot over {logical}, {logical}/2, ... threads; the host exposes {logical} logical CPUs)"}


def measured_traffic(n, B):
    """HBM bytes per launch of the solve kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, in bytes),
    as recorded by scripts/summarize_profile.py for this exact workload; None when no matching record is committed."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("n") == n and rec.get("batch") == B:
        return rec.get("bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU")
    ap.add_argument("--n", type=int, default=N_GRID)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-warm", action="store_true", help="skip the separately reported warm-start leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mpc_local_planner_amd as m

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    B, n = args.batch, args.n
    cfg = m.config_carlike_min_time(n=n, mu_init_warm=1e-2)      # only solves that are given an initial guess (the warm-start leg) use it
    solver = m.BatchSolver(cfg, max_batch=B, device=local_rank)
    # independent planner instances per rank: seed + rank (no scatter needed)
    from mpc_local_planner_amd import sharding
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=sharding.rank_seed(m.workloads.SEED_CONFIG2, rank))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dx0, dxf, dup, ddtp = T(x0), T(xf), T(up), T(dtp)
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev)
    uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)

    def step():
        solver.solve_device(B, dx0.data_ptr(), dxf.data_ptr(), dup.data_ptr(), ddtp.data_ptr(), None, None, None,
                            xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())

    def sync():
        solver.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        solver.synchronize()          # a control cycle ends when its commands are available
        kernel_ms.append(solver.last_kernel_ms())
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device=dev)

    status = st.cpu().numpy()
    iters = it.cpu().numpy()

    # ---- warm start, reported separately: the plant advances one controller period (0.2 s) with u_0, the previous
    #      solution is the initial guess with x0 overwritten (full_discretization_grid_base_se2.cpp:101-110; variable grid: no shifting)
    warm = None
    if world == 1 and not args.no_warm:
        per, Lw = 0.2, float(cfg.model_params[0])
        u0 = uo[:, 0, :].clone()
        x1 = dx0.clone()
        x1[:, 0] += per * u0[:, 0] * torch.cos(dx0[:, 2]); x1[:, 1] += per * u0[:, 0] * torch.sin(dx0[:, 2])
        x1[:, 2] = torch.remainder(dx0[:, 2] + per * u0[:, 0] * torch.tan(u0[:, 1]) / Lw + np.pi, 2 * np.pi) - np.pi
        xi, ui, di = xo.clone(), uo.clone(), do.clone()
        dper = torch.full((B,), per, dtype=torch.float64, device=dev)
        st2 = torch.empty_like(st); it2 = torch.empty_like(it)

        def wstep():
            solver.solve_device(B, x1.data_ptr(), dxf.data_ptr(), u0.data_ptr(), dper.data_ptr(), xi.data_ptr(), ui.data_ptr(), di.data_ptr(),
                                xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st2.data_ptr(), it2.data_ptr())
        for _ in range(args.warmup):
            wstep()
        sync()
        tw = time.perf_counter()
        for _ in range(args.steps):
            wstep()
            solver.synchronize()
        sync()
        tw = time.perf_counter() - tw
        was_ok = torch.from_numpy(status == 0).to(dev)
        s2, i2 = st2[was_ok].cpu().numpy(), it2[was_ok].cpu().numpy()
        warm = {"value": B * args.steps / tw, "unit": "solves/s", "ms_per_step": tw / args.steps * 1e3,
                "converged_frac_of_previously_converged": float((s2 == 0).mean()), "iters_mean": float(i2.mean()),
                "init": "previous solution with x0 advanced one 0.2 s period under u_0; slacks and multipliers re-initialised at mu0 = mu_init_warm = 1e-2"}
    if rank == 0:
        total = B * world * args.steps
        value = total / elapsed
        k_ms = float(np.mean(kernel_ms))
        bytes_per_launch = algorithmic_bytes_per_solve(n) * B
        achieved_gbs = bytes_per_launch / (k_ms * 1e-3) / 1e9
        mean_it = float(iters.mean())
        flops_per_iter = 914.0 * (n - 1)               # SURVEY.md 8d convention
        fp64_tf = B * mean_it * flops_per_iter / (k_ms * 1e-3) / 1e12
        line = {
            "metric": "MPC solves/sec (batched control cycles) at N=50 carlike",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: carlike (Ackermann) minimum-time MPC, n=50 grid points, "
                                   f"batch={B} instances per GPU, cold start (Controller::step on an empty grid), tol 1e-8, "
                                   "max 100 iterations", "n": n, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"instances sharded over {world} GPU(s), no data-path collective",
                       "seed": m.workloads.SEED_CONFIG2},
            "solver": {"converged_frac": float((status == 0).mean()), "iters_mean": mean_it,
                       "iters_p50": float(np.percentile(iters, 50)), "iters_p99": float(np.percentile(iters, 99)),
                       "iters_max": int(iters.max())},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": measured_traffic(n, B),
                         "kernel": "mpc_ipm_wave_kernel",
                         "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "latency/FP64-issue bound by construction (SURVEY.md 8d): compulsory traffic is ~4 KB per solve",
                         "fp64_valu": {"achieved_tflops": fp64_tf, "peak_tflops": FP64_VECTOR_PEAK_TF,
                                       "frac": fp64_tf / FP64_VECTOR_PEAK_TF,
                                       "flops_per_iteration": flops_per_iter}},
        }
        if warm is not None:
            line["warm_start"] = warm
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(line))
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
