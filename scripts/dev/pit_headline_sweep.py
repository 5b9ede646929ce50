"""developer tool: config 2, reference path alone, against the barrier value below which the serial sweeps take over (MPC_PIT_MU, -DMPC_DEV_SWITCHES build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpc_local_planner_amd as m
out = []
for seed in (None, 1, 2):
    B, n = 1024, 50
    inp = m.workloads.carlike_min_time_inputs(B) if seed is None else m.workloads.carlike_min_time_inputs(B, seed=seed)
    s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    r = s.solve(*inp); r = s.solve(*inp)
    out.append(f"seed {seed}: {int((r.status == 0).sum())}/{B} it {r.iters.mean():.2f} {s.last_kernel_ms():.3f} ms")
    s.close()
print("MPC_PIT_MU", os.environ.get("MPC_PIT_MU"), "MPC_NO_PIT", os.environ.get("MPC_NO_PIT"), "|", " | ".join(out))
