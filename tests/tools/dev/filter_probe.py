"""Developer probe (MI355X): the device's filter line search (mpc_config.line_search = MPC_LS_FILTER) against the C oracle's (oracle_config.line_search = 1) and against
the l1 merit, workload by workload: statuses, iteration counts, trajectories, converged shares and launch times.  Run from the repository root."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A
from oracle import se2_nlp as R, c_oracle as CO


def run(tag, cfg_fn, ocfg, inputs, B, obstacles=None, obst=None, **kw):
    out = {}
    for name, ls in (("merit", A.LS_MERIT), ("filter", A.LS_FILTER)):
        s = m.BatchSolver(cfg_fn(line_search=ls, **kw), max_batch=B)
        r = s.solve(*inputs, obstacles=obstacles)
        r = s.solve(*inputs, obstacles=obstacles)
        ms = s.last_kernel_ms()
        s.close()
        o = CO.solve_batch(CO.from_nlp_config(ocfg, line_search=0 if ls == A.LS_MERIT else 1), *inputs, obstacles=obstacles, obst=obst)
        both = (r.status == 0) & (o[3] == 0)
        err = np.abs(r.x - o[0]).reshape(B, -1).max(1)
        print(f"[{tag}] {name:6s}: device converged {(r.status == 0).mean():.4f} oracle {(o[3] == 0).mean():.4f} same status {(r.status == o[3]).mean():.4f} "
              f"same iters (both) {(r.iters == o[4])[both].mean():.3f} median err {np.median(err[both]):.1e} err<1e-6 {(err[both] < 1e-6).mean():.3f}  "
              f"mean iters dev {r.iters.mean():.1f} oracle {o[4].mean():.1f}  kernel {ms:.3f} ms", flush=True)
        out[name] = r
    return out


B = int(os.environ.get("B", 256))
run("config 2 n50", lambda **k: m.config_carlike_min_time(50, **k), R.config_carlike_min_time(50), m.workloads.carlike_min_time_inputs(B), B)
run("carlike n20", lambda **k: m.config_carlike_min_time(20, **k), R.config_carlike_min_time(20), m.workloads.carlike_min_time_inputs(B), B)
run("unicycle quad n20", lambda **k: m.config_unicycle_quadratic(20, **k), R.config_unicycle_quadratic(20), m.workloads.unicycle_quadratic_inputs(B), B)
run("bicycle n30", lambda **k: m.config_bicycle_min_time(30, **k), R.config_bicycle_min_time(30), m.workloads.bicycle_min_time_inputs(B, goal_range=(1.0, 5.0)), B)
run("config 5 n120", lambda **k: m.config_bicycle_min_time(120, **k), R.config_bicycle_min_time(120), m.workloads.bicycle_min_time_inputs(B), B)
# config 3 shape, rows binding
n, O, V, M = 80, 16, 6, 4
x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=(0.15, 0.8))
ocfg = R.config_unicycle_quadratic(n)
run("config 3 binding", lambda **k: m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M, **k), ocfg, (x0, xf, up, dtp), B, obstacles=obs,
    obst=CO.obst_from_nlp_config(ocfg, O, V, M))
# static points inside the clearance band (the restoration workload)
n, O = 30, 3
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=77, goal_range=(2.0, 4.0))
rng = np.random.default_rng(78)
d = xf[:, None, :2] - x0[:, None, :2]
nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, O, 1)) * d + rng.uniform(0.05, 0.5, (B, O, 1)) * rng.choice([-1.0, 1.0], (B, O, 1)) * nrm
obstacles = (np.full(B, O, np.int32), np.ones((B, O), np.int32), pts.reshape(B, O, 1, 2))
ocfg = R.config_carlike_min_time(n)
ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = 0.3, 0.5, 2.5
run("points in the band n30", lambda **k: m.config_carlike_min_time(n, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=O, max_vertices=1,
                                                                       max_obstacle_rows=4, **k), ocfg, (x0, xf, up, dtp), B, obstacles=obstacles, obst=CO.obst_from_nlp_config(ocfg, O, 1, 4))
