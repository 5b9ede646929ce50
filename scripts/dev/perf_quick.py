"""developer script (GPU): kernel times of the bench workloads in one short run (min over 10 launches), with convergence and iteration statistics"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m

def run(label, cfg, inp, B, reps=10, **kw):
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(*inp, **kw)
    ms = []
    for _ in range(reps):
        r = s.solve(*inp, **kw); ms.append(s.last_kernel_ms())
    w, tot = s.last_candidates(B)
    s.close()
    conv = np.mean(r.status == 0)
    print(f"{label}: kernel {min(ms):.3f} ms (median {np.median(ms):.3f})  {B * conv / min(ms):.1f}k conv solves/s  converged {conv:.4f}  iters {r.iters.mean():.2f}  all-candidate iters {np.mean(tot):.2f}  checksum {float(np.nansum(r.x[r.status == 0])):.9f}", flush=True)

C = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5))
CL = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 60, 50, 40), candidate_param=(0.0, 2.0, 3.0, 1.5))
run("config 2 headline B=1024 c4", m.config_carlike_min_time(50, **C), m.workloads.carlike_min_time_inputs(1024, seed=20260924), 1024)
run("config 2 one candidate B=1024", m.config_carlike_min_time(50), m.workloads.carlike_min_time_inputs(1024, seed=20260924), 1024)
run("config 4 share B=4096 c4 (large-batch caps)", m.config_carlike_min_time(50, **CL), m.workloads.carlike_min_time_inputs(4096, seed=20260924), 4096)
run("config 4 share B=4096 c4 (headline caps)", m.config_carlike_min_time(50, **C), m.workloads.carlike_min_time_inputs(4096, seed=20260924), 4096)
run("config 5 shape fp64 bicycle n=120 B=256", m.config_bicycle_min_time(120), m.workloads.bicycle_min_time_inputs(256), 256, reps=4)
