import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_local_planner_amd as m
g = np.load("tests/golden/unicycle_quadratic_obstacles_n80.npz")
B, O, V = g["x0"].shape[0], g["vertices"].shape[1], g["vertices"].shape[2]
s = m.BatchSolver(m.config_unicycle_quadratic(80, max_obstacles=O, max_vertices=V, max_obstacle_rows=int(g["max_rows"])), max_batch=B)
r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"]))
err = np.abs(r.x - g["x"]).reshape(B, -1).max(1)
print("status", r.status.tolist()); print("iters gpu", r.iters.tolist()); print("iters ref", g["iters"].tolist()); print("err", np.array2string(err, precision=2))
