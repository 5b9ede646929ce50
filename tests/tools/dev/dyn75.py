import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from oracle import se2_nlp as R, c_oracle as CO, kkt_check as KC
import mpc_local_planner_amd.workloads as W
from test_gpu_ext_rows import point_obstacles, FOOTPRINTS
B, n = 128, 50
kind, params, dmin = FOOTPRINTS["line"]
x0, xf, up, dtp = W.carlike_min_time_inputs(B, seed=931, goal_range=(2.0, 5.0))
no, nv, vt = point_obstacles(x0, xf, 932, n_obst=3, lo=0.6, hi=1.1)
rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
d = xf[:, :2] - x0[:, :2]
nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.2 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
ocfg = R.config_carlike_min_time(n)
ocfg.footprint_kind, ocfg.footprint_params = kind, params
ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, dmin, 0.5, 2.5
CO.build()
ref = CO.solve_batch(CO.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel), obst=CO.obst_from_nlp_config(ocfg, 3, 1, 4))
I = int(sys.argv[1]) if len(sys.argv) > 1 else 75
print("oracle: status", ref[3][I], "iters", ref[4][I], "dt", ref[2][I])
res = KC.kkt_many(ocfg, x0, xf, up, dtp, ref[0], ref[1], ref[2], [I], obstacles=(no, nv, vt, rad, vel), max_rows=4)
print("oracle answer through the checker:", res[I])
try:
    import torch
    if torch.cuda.is_available():
        import mpc_local_planner_amd as m
        s = m.BatchSolver(m.config_carlike_min_time(n, footprint_kind=kind, footprint_params=params, enable_dynamic_obstacles=True, min_obstacle_dist=dmin,
                                                    force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4), max_batch=B)
        r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel))
        print("device: status", r.status[I], "iters", r.iters[I], "dt", r.dt[I], "|x - oracle|", np.abs(r.x[I] - ref[0][I]).max())
        res = KC.kkt_many(ocfg, x0, xf, up, dtp, r.x, r.u, r.dt, [I], obstacles=(no, nv, vt, rad, vel), max_rows=4)
        print("device answer through the checker:", res[I])
        diff = np.flatnonzero((r.status != ref[3]) | (np.abs(r.iters - ref[4]) > 0))
        print("instances with different status / iterations:", [(int(i), int(r.status[i]), int(ref[3][i]), int(r.iters[i]), int(ref[4][i])) for i in diff])
except ImportError:
    pass
CO._load().oracle_set_trace(C.c_int(1))
sl = slice(I, I + 1)
CO.solve_batch(CO.from_nlp_config(ocfg), x0[sl], xf[sl], up[sl], dtp[sl], obstacles=(no[sl], nv[sl], vt[sl], rad[sl], vel[sl]), obst=CO.obst_from_nlp_config(ocfg, 3, 1, 4), nthreads=1)
