"""developer script (GPU): kernel times of the global-form workloads with the library named by MPC_HIP_LIB (a one-model developer build: MODEL env = 0 unicycle / 3 bicycle)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
def run(label, cfg, inp, B, reps=4, obstacles=None):
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(*inp, obstacles=obstacles); ms = []
    for _ in range(reps):
        r = s.solve(*inp, obstacles=obstacles); ms.append(s.last_kernel_ms())
    wg, lds = s.occupancy(B)
    s.close()
    print(f"{label:44s} kernel {min(ms):8.3f} ms (median {np.median(ms):8.3f})  converged {np.mean(r.status == 0):.4f}  iterations {r.iters.mean():.2f}  wg/CU {wg} lds {lds}  checksum {float(np.nansum(r.x[r.status == 0])):.6f}", flush=True)
print("library:", os.environ.get("MPC_HIP_LIB", "product"))
if os.environ.get("MODEL", "3") == "3":
    C5 = dict(candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0))
    run("config5 fp64 bicycle n120 B1024 c4", m.config_bicycle_min_time(120, **C5), m.workloads.bicycle_min_time_inputs(1024), 1024)
    run("config5 fp64 bicycle n120 B4096 single", m.config_bicycle_min_time(120), m.workloads.bicycle_min_time_inputs(4096), 4096)
    run("config5 fp64 bicycle n120 B8192 c4", m.config_bicycle_min_time(120, **C5), m.workloads.bicycle_min_time_inputs(8192), 8192, reps=2)
else:
    n, O, V, M = 80, 16, 6, 4
    for lat in ((0.15, 0.8), (0.3, 1.5)):
        x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(4096, n_obst=O, max_vertices=V, lateral=lat)
        run(f"config3 n80 B4096 lateral {lat}", m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), (x0, xf, up, dtp), 4096, obstacles=obs)
