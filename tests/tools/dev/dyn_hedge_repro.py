"""developer script (GPU): the hedged dynamic-obstacle + line-footprint workload of tests/test_gpu_ext_rows.py; KKT check of the instances a hedge answers
(oracle/kkt_check.py is the checker only) and of the same candidates solved as (capped reference, that candidate) pairs"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
from oracle import se2_nlp as R
from test_gpu_ext_rows import point_obstacles, FOOTPRINTS
name = sys.argv[1] if len(sys.argv) > 1 else "line"
B, n = 128, 50
kind, params, dmin = FOOTPRINTS[name]
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=931, goal_range=(2.0, 5.0))
no, nv, vt = point_obstacles(x0, xf, 932, n_obst=3, lo=0.6, hi=1.1)
rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
d = xf[:, :2] - x0[:, :2]
nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.2 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
ocfg = R.config_carlike_min_time(n)
ocfg.footprint_kind, ocfg.footprint_params = kind, params
ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, dmin, 0.5, 2.5
base = dict(footprint_kind=kind, footprint_params=params, enable_dynamic_obstacles=True, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4)
obs = (no, nv, vt, rad, vel)
sc = m.BatchSolver(m.config_carlike_min_time(n, candidates=(0, 5, 5, 6), candidate_max_iter=(100, 100, 100, 100), candidate_param=(0.0, 2.0, 1.0, 2.0), **base), max_batch=B)
rc = sc.solve(x0, xf, up, dtp, obstacles=obs)
wc, tot = sc.last_candidates(B)
sc.close()
out = dict(x=rc.x, u=rc.u, dt=rc.dt, status=rc.status, iters=rc.iters, win=wc, tot=tot)
for c, (k, p) in enumerate(((5, 2.0), (5, 1.0), (6, 2.0)), start=1):
    s1 = m.BatchSolver(m.config_carlike_min_time(n, candidates=(0, k), candidate_max_iter=(1, 100), candidate_param=(0.0, p), **base), max_batch=B)
    r1 = s1.solve(x0, xf, up, dtp, obstacles=obs)
    w1, _ = s1.last_candidates(B)
    s1.close()
    out.update({f"x{c}": r1.x, f"u{c}": r1.u, f"dt{c}": r1.dt, f"status{c}": r1.status, f"iters{c}": r1.iters, f"win{c}": w1})
np.savez(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03", f"dyn_repro_{name}{os.environ.get('REPRO_TAG', '')}.npz"), **out)
print("saved", {k: v.shape for k, v in out.items() if k in ("x", "win")}, "converged", float(np.mean(rc.status == 0)), "winners", np.bincount(wc + 1, minlength=5).tolist())
