import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_local_planner_amd as m
n, B = 50, 1024
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
for mu in (0.1, 1e-2, 1e-3, 1e-4):
    cold = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    r = cold.solve(x0, xf, up, dtp)
    ok = r.status == 0
    # advance the plant one controller period (0.2 s) with u0 (car-like model, explicit Euler), then re-solve warm
    L, per = 0.4, 0.2
    x1 = x0.copy()
    u0 = r.u[:, 0]
    x1[:, 0] += per * u0[:, 0] * np.cos(x0[:, 2]); x1[:, 1] += per * u0[:, 0] * np.sin(x0[:, 2])
    x1[:, 2] = (x0[:, 2] + per * u0[:, 0] * np.tan(u0[:, 1]) / L + np.pi) % (2 * np.pi) - np.pi
    warm = m.BatchSolver(m.config_carlike_min_time(n, mu_init=mu), max_batch=B)
    w = warm.solve(x1[ok], xf[ok], u0[ok], np.full(ok.sum(), per), init=(r.x[ok], r.u[ok], r.dt[ok]))
    print("mu_init %.0e: cold conv %.3f iters mean %.1f | warm conv %.3f iters mean %.1f p50 %.0f p90 %.0f max %d kernel ms %.2f | dT mean %.3f" % (
        mu, ok.mean(), r.iters.mean(), (w.status == 0).mean(), w.iters.mean(), np.median(w.iters), np.percentile(w.iters, 90), w.iters.max(), warm.last_kernel_ms(),
        ((w.dt - r.dt[ok]) * (n - 1))[w.status == 0].mean()))
