"""developer tool: partitioned sweeps in the kernels with clearance rows, against the barrier value below which the serial sweeps take over (MPC_PIT_MU, -DMPC_PIT_OBST -DMPC_DEV_SWITCHES build)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import mpc_local_planner_amd as m
from test_gpu_ext_rows import FOOTPRINTS, point_obstacles
out = []
for name in ("line", "polygon", "two_circles"):
    B, n = 192, 50
    kind, params, dmin = FOOTPRINTS[name]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=901, goal_range=(2.0, 5.0))
    no, nv, vt = point_obstacles(x0, xf, 902)
    kw = dict(footprint_kind=kind, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=4, max_vertices=1, max_obstacle_rows=4)
    kw.update(dict(footprint_vertices=params) if kind == 4 else dict(footprint_params=params))
    s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    out.append(f"{name} {int((r.status == 0).sum())}/{B} it {r.iters.mean():.1f} {s.last_kernel_ms():.2f} ms")
    s.close()
for lat in ((0.15, 0.8), (0.3, 1.5)):
    B, n, O, V, M = 4096, 80, 16, 6, 4
    x0, xf, up, dtp, ob = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=lat)
    s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M, max_iter=60), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=ob); r = s.solve(x0, xf, up, dtp, obstacles=ob)
    out.append(f"config3 {lat} {int((r.status == 0).sum())}/{B} {s.last_kernel_ms():.2f} ms")
    s.close()
print("MPC_PIT_MU", os.environ.get("MPC_PIT_MU"), "MPC_NO_PIT", os.environ.get("MPC_NO_PIT"), "|", " | ".join(out))
