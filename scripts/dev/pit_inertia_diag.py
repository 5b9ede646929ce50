import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import mpc_local_planner_amd as m
from test_gpu_ext_rows import FOOTPRINTS, point_obstacles
name = sys.argv[1] if len(sys.argv) > 1 else "polygon"
B, n = 64, 50
kind, params, dmin = FOOTPRINTS[name]
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(192, seed=901, goal_range=(2.0, 5.0))
no, nv, vt = point_obstacles(x0, xf, 902)
sl = slice(0, B)
kw = dict(footprint_kind=kind, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=4, max_vertices=1, max_obstacle_rows=4)
kw.update(dict(footprint_vertices=params) if kind == 4 else dict(footprint_params=params))
s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
r = s.solve(x0[sl], xf[sl], up[sl], dtp[sl], obstacles=(no[sl], nv[sl], vt[sl]))
print("converged", (r.status == 0).sum(), "of", B, "iters", r.iters.mean())
s.close()
