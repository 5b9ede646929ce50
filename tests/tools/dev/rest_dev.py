import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import torch
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A
from oracle import c_oracle as CO, se2_nlp as R
CO.build()
B, n, O = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 30, 3
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=77, goal_range=(2.0, 4.0))
rng = np.random.default_rng(78)
d = xf[:, None, :2] - x0[:, None, :2]
nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, O, 1)) * d + rng.uniform(0.05, 0.5, (B, O, 1)) * rng.choice([-1.0, 1.0], (B, O, 1)) * nrm
obstacles = (np.full(B, O, np.int32), np.ones((B, O), np.int32), pts.reshape(B, O, 1, 2))
ocfg = R.config_carlike_min_time(n)
ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = 0.3, 0.5, 2.5
ob = CO.obst_from_nlp_config(ocfg, O, 1, 4)
if os.environ.get('NO_RESTORATION'): CO._load().oracle_set_algo(C.c_int(10), C.c_double(0.0))
on = CO.solve_batch(CO.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=obstacles, obst=ob)
s = m.BatchSolver(m.config_carlike_min_time(n, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=O, max_vertices=1, max_obstacle_rows=4), max_batch=B)
r = s.solve(x0, xf, up, dtp, obstacles=obstacles)
s.close()
diff = np.flatnonzero((r.status != on[3]) | (r.iters != on[4]))
print("n", n, "device converged", (r.status == 0).sum(), "oracle", (on[3] == 0).sum(), "differing:", [(int(i), int(r.status[i]), int(r.iters[i]), int(on[3][i]), int(on[4][i])) for i in diff])
