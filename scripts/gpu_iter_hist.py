import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_local_planner_amd as m
B, n = 4096, 50
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
r = s.solve(x0, xf, up, dtp)
ok = r.status == 0
print("converged", ok.mean(), "status counts", np.bincount(r.status, minlength=5))
print("iters of converged: percentiles 50/90/99/99.9/max", np.percentile(r.iters[ok], [50, 90, 99, 99.9, 100]))
print("hist converged (bins of 10):", np.histogram(r.iters[ok], bins=np.arange(0, 111, 10))[0])
print("hist failed    (bins of 10):", np.histogram(r.iters[~ok], bins=np.arange(0, 111, 10))[0])
