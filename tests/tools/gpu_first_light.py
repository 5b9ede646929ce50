"""First-light GPU check: parity vs the numpy oracle on a few instances + B=1024 timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpc_local_planner_amd as m
from oracle import se2_nlp as R, ipm_dense as I

n = 50
cfg = m.config_carlike_min_time(n=n)
B = 1024
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
solver = m.BatchSolver(cfg, max_batch=B, device=0)
t = time.time(); res = solver.solve(x0, xf, up, dtp); t1 = time.time() - t
print("first solve (incl. copies) s:", t1, "kernel ms:", solver.last_kernel_ms())
for rep in range(3):
    t = time.time(); res = solver.solve(x0, xf, up, dtp); t1 = time.time() - t
    print("solve s:", t1, "kernel ms:", solver.last_kernel_ms())
st, it = res.status, res.iters
print("status counts:", {k: int((st == k).sum()) for k in range(5)})
print("iters: mean %.1f median %.0f p90 %.0f p99 %.0f max %d" % (it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max()))
ms = solver.last_kernel_ms()
print("throughput solves/s (kernel):", B / (ms * 1e-3))
ocfg = R.config_carlike_min_time(n)
worst = 0
for i in range(6):
    inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
    ref = I.solve(ocfg, inp, R.cold_start(ocfg, x0[i], xf[i]), opt=I.IpmOptions(max_iter=100))
    err = max(np.abs(ref.traj.x - res.x[i]).max(), np.abs(ref.traj.u - res.u[i, :-1]).max(), abs(ref.traj.dt - res.dt[i]))
    print(i, "oracle", ref.status, ref.iters, "gpu", st[i], it[i], "err", err)
