#!/usr/bin/env python
"""Developer tool (GPU box): what the arithmetic precision buys on the headline workload (config 2 with the bench candidate set) and on
config 3 (unicycle n = 80, 16 polygons): kernel time, converged fraction and distance to the fp64 result for MPC_FP64 / MPC_MIXED / MPC_FP32.
Output of the round-2 run: profiles/r02_precision_check.log."""
import sys, os, numpy as np, torch
torch.zeros(1, device="cuda")          # torch initialises HIP first
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_local_planner_amd as m

print("== config 2 (car-like minimum time, n = 50), candidates (0,5,5,7) caps (60,45,40,35); prec 0 = fp64, 2 = mixed, 1 = fp32 (tol 1e-4)")
n = 50
K, CAPS, PAR = (0, 5, 5, 7), (60, 45, 40, 35), (0, 2.0, 3.0, 1.5)
for B in (1024, 4096):
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    ref = None
    for prec in (0, 2, 1):
        s = m.BatchSolver(m.config_carlike_min_time(n, precision=prec, tol=(1e-4 if prec == 1 else 1e-8), candidates=K, candidate_max_iter=CAPS, candidate_param=PAR), max_batch=B)
        r = s.solve(x0, xf, up, dtp); r = s.solve(x0, xf, up, dtp)
        ok = r.status == 0
        ms = s.last_kernel_ms()
        print("B", B, "prec", prec, "kernel ms %.2f" % ms, "iters mean %.1f" % r.iters.mean(), "conv %.4f" % ok.mean(), "conv solves/s %.0fk" % (B * ok.mean() / ms), flush=True)
        if ref is None: ref = r
        elif ok.any():
            both = ok & (ref.status == 0)
            d = np.abs(r.x - ref.x).reshape(B, -1).max(1)
            print("   vs fp64: median %.2e p95 %.2e  frac<1e-4 %.4f frac<1e-6 %.4f" % (np.median(d[both]), np.percentile(d[both], 95), (d[both] < 1e-4).mean(), (d[both] < 1e-6).mean()))
        s.close()

print("== config 3 (unicycle quadratic form, n = 80, 16 polygons, B = 4096), single candidate; prec 0 = fp64, 1 = fp32")
n, O, V, M, B = 80, 16, 6, 4, 4096
x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
ref = None
for prec, tol in ((0, None), (1, 1e-4), (1, 1e-3)):
    kw = dict(precision=prec)
    if tol: kw["tol"] = tol
    s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M, **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=obs); r = s.solve(x0, xf, up, dtp, obstacles=obs)
    ok = r.status == 0
    print("prec", prec, "tol", tol, "kernel ms %.2f" % s.last_kernel_ms(), "iters mean %.1f" % r.iters.mean(), "conv %.4f" % ok.mean(), flush=True)
    if ref is None: ref = r
    else:
        both = ok & (ref.status == 0)
        d = np.abs(r.x - ref.x).reshape(B, -1).max(1)
        print("   vs fp64: median %.2e p95 %.2e max %.2e  frac<1e-3 %.4f" % (np.median(d[both]), np.percentile(d[both], 95), d[both].max(), (d[both] < 1e-3).mean()))
    s.close()
