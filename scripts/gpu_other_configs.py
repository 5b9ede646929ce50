"""Sanity + timing of the other BASELINE.json configs on one GPU (parity for them is in tests/test_gpu_parity.py at smaller sizes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpc_local_planner_amd as m

def run(name, solver, args, kw={}):
    r = solver.solve(*args, **kw)
    t = time.perf_counter(); r = solver.solve(*args, **kw); wall = time.perf_counter() - t
    B = args[0].shape[0]
    print(f"{name}: B={B} kernel {solver.last_kernel_ms():.2f} ms  ({B / solver.last_kernel_ms() * 1e3:.0f} solves/s device-resident, {B / wall:.0f} incl. PCIe/host) "
          f"converged {np.mean(r.status == 0):.3f} iters mean {r.iters.mean():.1f} p99 {np.percentile(r.iters, 99):.0f}")
    return r

# config 3: unicycle quadratic, n=80, 16 polygon obstacles, batch 4096
n, B, O, V, M = 80, 4096, 16, 6, 4
x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), max_batch=B)
run("config 3 (unicycle quadratic n=80, 16 polygons)", s, (x0, xf, up, dtp), dict(obstacles=(no, nv, verts)))
s.close()
# config 4 per-GPU share: car-like n=50, batch 4096
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(4096)
s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=4096)
run("config 4 share (car-like n=50, B=4096 per GPU)", s, (x0, xf, up, dtp))
s.close()
# config 5 per-GPU share: bicycle n=120 fp32, batch 1024
x0, xf, up, dtp = m.workloads.bicycle_min_time_inputs(1024)
for prec, tol in ((1, 1e-4), (0, 1e-8)):
    s = m.BatchSolver(m.config_bicycle_min_time(120, precision=prec, tol=tol), max_batch=1024)
    run(f"config 5 share (bicycle n=120, {'fp32' if prec else 'fp64'}, B=1024 per GPU)", s, (x0, xf, up, dtp))
    s.close()
