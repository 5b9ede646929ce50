import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_local_planner_amd as m
B, n = 1024, 50
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
ref = s.solve(x0, xf, up, dtp)
for rep in range(8):
    r = s.solve(x0, xf, up, dtp)
    dif = np.nonzero((r.x != ref.x).reshape(B, -1).any(1) | (r.iters != ref.iters))[0]
    print(rep, "differing instances:", len(dif), dif[:10], "iters", r.iters[dif[:5]], ref.iters[dif[:5]], "maxdiff", np.abs(r.x - ref.x).max())
