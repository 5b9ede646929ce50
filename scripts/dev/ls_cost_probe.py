"""Developer probe (MI355X): launch time per solver iteration under the two line searches (mpc_config.line_search), reference path alone.  No oracle involved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A

def run(tag, mk, inputs, B, **kw):
    for name, ls in (("merit", A.LS_MERIT), ("filter", A.LS_FILTER)):
        s = m.BatchSolver(mk(line_search=ls, **kw), max_batch=B)
        ms = []
        for _ in range(4):
            r = s.solve(*inputs); ms.append(s.last_kernel_ms())
        s.close()
        it = r.iters.astype(np.int64)
        print(f"[{tag}] {name:6s} kernel {min(ms):8.3f} ms  converged {(r.status == 0).mean():.4f}  iterations sum {it.sum()} mean {it.mean():.2f} max {it.max()}  ns per iteration-instance {1e6 * min(ms) / it.sum():.2f}", flush=True)

for B in (1024, 4096):
    run(f"car-like n50 B={B}", lambda **k: m.config_carlike_min_time(50, **k), m.workloads.carlike_min_time_inputs(B), B)
    run(f"car-like n20 B={B}", lambda **k: m.config_carlike_min_time(20, **k), m.workloads.carlike_min_time_inputs(B, goal_range=(1.0, 2.4)), B)
    run(f"bicycle n120 B={B}", lambda **k: m.config_bicycle_min_time(120, **k), m.workloads.bicycle_min_time_inputs(B), B)
run("car-like n20 B=32768", lambda **k: m.config_carlike_min_time(20, **k), m.workloads.carlike_min_time_inputs(32768, goal_range=(1.0, 2.4)), 32768)
