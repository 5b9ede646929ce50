"""Batch-scale device-vs-C-oracle parity for the rows that live in the EXT kernel instantiation (terminal ball, integral form with dt free,
line / polygon / two-circle footprints, dynamic obstacles).  Written after round 1's GPU minutes were spent: run it on an MI355X
(`gpurun -- python tests/tools/gpu_batch_parity_ext_rows.py`) and turn the printed statistics into `-m gpu` tests with thresholds.
The C oracle is pinned to the numpy fixtures for every one of these rows (tests/test_oracle_solver.py)."""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import workloads as W


def report(name, r, ref):
    xo, uo, do, st, it = ref
    B = xo.shape[0]
    both = (r.status == 0) & (st == 0)
    err = np.maximum(np.abs(r.x - xo).reshape(B, -1).max(1), np.abs(r.u - uo).reshape(B, -1).max(1))
    print("%-28s B %4d  conv gpu %.3f oracle %.3f same-status %.3f | err median %.1e p90 %.1e  <1e-4: %.3f | iters gpu %.1f oracle %.1f |d|<=2: %.3f" % (
        name, B, (r.status == 0).mean(), (st == 0).mean(), (r.status == st).mean(), np.median(err[both]), np.percentile(err[both], 90),
        (err[both] < 1e-4).mean(), r.iters.mean(), it.mean(), (np.abs(r.iters - it)[both] <= 2).mean()), flush=True)


def point_obstacles(x0, xf, seed, n_obst=4, lo=0.3, hi=0.9):
    rng = np.random.default_rng(seed)
    B = x0.shape[0]
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, n_obst, 1)) * d + rng.uniform(lo, hi, (B, n_obst, 1)) * rng.choice([-1.0, 1.0], (B, n_obst, 1)) * nrm
    return np.full(B, n_obst, np.int32), np.ones((B, n_obst), np.int32), pts.reshape(B, n_obst, 1, 2)


B, n = 256, 50
x0, xf, up, dtp = W.carlike_min_time_inputs(B, seed=901, goal_range=(2.0, 5.0))
POLY = (0.25, -0.05, 0.18, -0.05, 0.18, -0.18, -0.19, -0.18, -0.25, 0.0, -0.19, 0.18, 0.18, 0.18, 0.18, 0.05, 0.25, 0.05)
for name, kind, params, dmin in (("line footprint", 2, (0.0, 0.0, 0.4, 0.0), 0.27), ("polygon footprint", 4, POLY, 0.15), ("two circles", 3, (0.2, 0.15, 0.2, 0.15), 0.1)):
    no, nv, vt = point_obstacles(x0, xf, 902)
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = kind, params, dmin, 0.5, 2.5
    kw = dict(footprint_kind=kind, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=4, max_vertices=1, max_obstacle_rows=4)
    kw.update(dict(footprint_vertices=params) if kind == 4 else dict(footprint_params=params))
    s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    report(name, r, CO.solve_batch(CO.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt), obst=CO.obst_from_nlp_config(ocfg, 4, 1, 4)))
    s.close()

# dynamic obstacles: one moving circle crossing the path + static points
no, nv, vt = point_obstacles(x0, xf, 903, n_obst=3, lo=0.5, hi=1.0)
rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
d = xf[:, :2] - x0[:, :2]
nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.0 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
ocfg = R.config_carlike_min_time(n)
ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, 0.3, 0.5, 2.5
s = m.BatchSolver(m.config_carlike_min_time(n, enable_dynamic_obstacles=True, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5,
                                            max_obstacles=3, max_vertices=1, max_obstacle_rows=4), max_batch=B)
r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel))
report("dynamic obstacles", r, CO.solve_batch(CO.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel), obst=CO.obst_from_nlp_config(ocfg, 3, 1, 4)))
s.close()

# terminal ball (effort-dominated quadratic form, goal within reach) and integral form with dt free
xq0, xqf, uq, dq = W.unicycle_quadratic_inputs(B, seed=904, goal_range=(0.8, 1.3))
ocfg = R.config_unicycle_quadratic(20)
ocfg.Q, ocfg.R, ocfg.Qf, ocfg.terminal_ball_S, ocfg.terminal_ball_gamma = np.array([0.2, 0.2, 0.02]), np.array([1.0, 0.5]), None, np.array([1.0, 1.0, 0.01]), 0.02
s = m.BatchSolver(m.config_unicycle_quadratic(20, Q=(0.2, 0.2, 0.02), R=(1.0, 0.5), Qf=None, terminal_ball_S=(1.0, 1.0, 0.01), terminal_ball_gamma=0.02), max_batch=B)
report("terminal ball", s.solve(xq0, xqf, uq, dq), CO.solve_batch(CO.from_nlp_config(ocfg), xq0, xqf, uq, dq))
s.close()
xq0, xqf, uq, dq = W.unicycle_quadratic_inputs(B, seed=905, goal_range=(1.0, 2.0))
ocfg = R.config_unicycle_quadratic(20)
ocfg.dt_free, ocfg.dt_lb, ocfg.dt_ub, ocfg.xf_fixed, ocfg.Qf, ocfg.integral_form, ocfg.R = True, 0.01, 2.0, (True, True, True), None, True, np.array([1.0, 0.5])
s = m.BatchSolver(m.config_unicycle_quadratic(20, dt_free=True, dt_lb=0.01, dt_ub=2.0, xf_fixed=(True, True, True), Qf=None, integral_form=True, R=(1.0, 0.5)), max_batch=B)
report("integral form, dt free", s.solve(xq0, xqf, uq, dq), CO.solve_batch(CO.from_nlp_config(ocfg), xq0, xqf, uq, dq))
s.close()
