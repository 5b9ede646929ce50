import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_local_planner_amd as m
g = np.load("tests/golden/unicycle_quadratic_n20.npz")
for tol, mi in [(1e-8, 100), (1e-6, 100), (1e-4, 100), (1e-8, 400)]:
    cfg = m.config_unicycle_quadratic(20, tol=tol, max_iter=mi)
    s = m.BatchSolver(cfg, max_batch=8)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    print(tol, mi, r.status, r.iters, np.abs(r.x - g["x"]).max(), np.abs(r.u - g["u"]).max())
    s.close()
