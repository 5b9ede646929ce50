import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpc_local_planner_amd as m
from oracle import c_oracle as CO, se2_nlp as R
n, B, O, V, M = 80, 256, 16, 6, 4
x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), max_batch=B)
r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
cfg = R.config_unicycle_quadratic(n)
oc = CO.from_nlp_config(cfg); ob = CO.obst_from_nlp_config(cfg, O, V, M)
xo, uo, do, st, it = CO.solve_batch(oc, x0, xf, up, dtp, obstacles=(no, nv, verts), obst=ob)
print("gpu conv", (r.status == 0).mean(), "status hist", np.bincount(r.status, minlength=5), "iters mean", r.iters.mean())
print("C   conv", (st == 0).mean(), "status hist", np.bincount(st, minlength=5), "iters mean", it.mean())
both = (r.status == 0) & (st == 0)
err = np.abs(r.x - xo).reshape(B, -1).max(1)
print("both converged:", both.sum(), "median err", np.median(err[both]), "max err", err[both].max())
bad = np.nonzero((r.status != 0) & (st == 0))[0][:10]
print("gpu-failed / C-ok instances:", bad.tolist(), "gpu status", r.status[bad].tolist(), "gpu iters", r.iters[bad].tolist(), "C iters", it[bad].tolist())
