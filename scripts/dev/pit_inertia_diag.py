"""developer tool: run one batch on a -DMPC_PIT_CHECK=64 build (MPC_HIP_LIB): the kernel prints every factorisation on which the partitioned and the serial sweep differ
(verdict on the inertia, or the step beyond 1e-7 relative)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpc_local_planner_amd as m
B, n = 64, 50
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
r = s.solve(x0, xf, up, dtp)
print("converged", (r.status == 0).sum(), "of", B, "iters", r.iters.mean())
s.close()
