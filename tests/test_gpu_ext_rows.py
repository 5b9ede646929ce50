"""GPU parity at batch scale (-m gpu) for the rows of the EXT kernel instantiation -- line / polygon / two-circle footprints, dynamic
obstacles, terminal ball, integral form with dt free -- plus the front-wheel car and the host-pointer staging path.  Every batch is
B >= 128 instances against the C oracle (oracle/mpc_oracle.c, pinned to the numpy fixtures in tests/test_oracle_solver.py) with the
accounting of tests/_parity.py: match / other KKT point of the reference-form NLP / unclassified (must be 0)."""
import time

import numpy as np
import pytest

from _parity import account

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("these tests need the MI355X (no HIP device here)")
    torch.zeros(1, device="cuda")
    import mpc_local_planner_amd as pkg
    return pkg


def point_obstacles(x0, xf, seed, n_obst=4, lo=0.3, hi=0.9):
    rng = np.random.default_rng(seed)
    B = x0.shape[0]
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, n_obst, 1)) * d + rng.uniform(lo, hi, (B, n_obst, 1)) * rng.choice([-1.0, 1.0], (B, n_obst, 1)) * nrm
    return np.full(B, n_obst, np.int32), np.ones((B, n_obst), np.int32), pts.reshape(B, n_obst, 1, 2)


POLY = (0.25, -0.05, 0.18, -0.05, 0.18, -0.18, -0.19, -0.18, -0.25, 0.0, -0.19, 0.18, 0.18, 0.18, 0.18, 0.05, 0.25, 0.05)   # the car-like example YAML's vertex list
FOOTPRINTS = {"line": (2, (0.0, 0.0, 0.4, 0.0), 0.27), "polygon": (4, POLY, 0.15), "two_circles": (3, (0.2, 0.15, 0.2, 0.15), 0.1)}


def _summary(r, ref, B, min_frac=0.5):
    both = (r.status == 0) & (ref[3] == 0)
    assert both.sum() >= min_frac * B, "too few instances converge on both sides to say anything"
    assert abs(int((r.status == 0).sum()) - int((ref[3] == 0).sum())) <= 0.08 * B
    err = np.abs(r.x - ref[0]).reshape(B, -1).max(1)
    assert np.median(err[both]) < 1e-7
    return both


@pytest.mark.parametrize("name", sorted(FOOTPRINTS))
def test_heading_dependent_footprints_batch_vs_c_oracle(m, c_oracle, name):
    """a21: line / polygon / two-circle footprints (teb footprint models) against point obstacles beside the path, car-like n = 50."""
    from oracle import se2_nlp as R
    B, n = 192, 50
    kind, params, dmin = FOOTPRINTS[name]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=901, goal_range=(2.0, 5.0))
    no, nv, vt = point_obstacles(x0, xf, 902)
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = kind, params, dmin, 0.5, 2.5
    kw = dict(footprint_kind=kind, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=4, max_vertices=1, max_obstacle_rows=4)
    kw.update(dict(footprint_vertices=params) if kind == 4 else dict(footprint_params=params))
    s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt), obst=c_oracle.obst_from_nlp_config(ocfg, 4, 1, 4))
    _summary(r, ref, B)
    account(f"{name} footprint, B={B}", ocfg, (x0, xf, up, dtp), r, ref, obstacles=(no, nv, vt), max_rows=4)
    s.close()


@pytest.mark.parametrize("ls", ["filter", "merit"])
def test_dynamic_obstacles_batch_vs_c_oracle(m, c_oracle, ls):
    """a22: one moving circle crossing the path + static points, car-like n = 50 (rows at t = k dt, dt in gradient and Hessian); under the default line search (filter) and
    under MPC_LS_MERIT (the extended kernel level's l1-merit path)."""
    from oracle import se2_nlp as R
    from mpc_local_planner_amd import _abi as A
    B, n = 192, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=901, goal_range=(2.0, 5.0))
    no, nv, vt = point_obstacles(x0, xf, 903, n_obst=3, lo=0.5, hi=1.0)
    rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
    d = xf[:, :2] - x0[:, :2]
    nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.0 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
    ocfg = R.config_carlike_min_time(n)
    ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, 0.3, 0.5, 2.5
    s = m.BatchSolver(m.config_carlike_min_time(n, enable_dynamic_obstacles=True, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5,
                                                max_obstacles=3, max_vertices=1, max_obstacle_rows=4, line_search=A.LS_MERIT if ls == "merit" else A.LS_DEFAULT), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel))
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, line_search=0 if ls == "merit" else 1), x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel), obst=c_oracle.obst_from_nlp_config(ocfg, 3, 1, 4))
    _summary(r, ref, B)
    account(f"dynamic obstacles, {ls} line search, B={B}", ocfg, (x0, xf, up, dtp), r, ref, obstacles=(no, nv, vt, rad, vel), max_rows=4)
    s.close()


def test_terminal_ball_batch_vs_c_oracle(m, c_oracle):
    from oracle import se2_nlp as R
    B = 192
    x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(B, seed=904, goal_range=(0.8, 1.3))
    ocfg = R.config_unicycle_quadratic(20)
    ocfg.Q, ocfg.R, ocfg.Qf, ocfg.terminal_ball_S, ocfg.terminal_ball_gamma = np.array([0.2, 0.2, 0.02]), np.array([1.0, 0.5]), None, np.array([1.0, 1.0, 0.01]), 0.02
    s = m.BatchSolver(m.config_unicycle_quadratic(20, Q=(0.2, 0.2, 0.02), R=(1.0, 0.5), Qf=None, terminal_ball_S=(1.0, 1.0, 0.01), terminal_ball_gamma=0.02), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    _summary(r, ref, B)
    account(f"terminal ball, B={B}", ocfg, (x0, xf, up, dtp), r, ref)
    s.close()


def test_integral_form_free_dt_batch_vs_c_oracle(m, c_oracle):
    from oracle import se2_nlp as R
    B = 192
    x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(B, seed=905, goal_range=(1.0, 2.0))
    ocfg = R.config_unicycle_quadratic(20)
    ocfg.dt_free, ocfg.dt_lb, ocfg.dt_ub, ocfg.xf_fixed, ocfg.Qf, ocfg.integral_form, ocfg.R = True, 0.01, 2.0, (True, True, True), None, True, np.array([1.0, 0.5])
    s = m.BatchSolver(m.config_unicycle_quadratic(20, dt_free=True, dt_lb=0.01, dt_ub=2.0, xf_fixed=(True, True, True), Qf=None, integral_form=True, R=(1.0, 0.5)), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    _summary(r, ref, B)
    account(f"integral form, dt free, B={B}", ocfg, (x0, xf, up, dtp), r, ref)
    s.close()


def test_front_wheel_car_batch_vs_c_oracle(m, c_oracle):
    """a13 front-wheel variant (simple_car.h:131-141: theta' = v sin(phi) / L), car-like min-time n = 50, against the C oracle."""
    import copy
    from oracle import se2_nlp as R
    from mpc_local_planner_amd import _abi as A
    B, n = 192, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=911)
    ocfg = copy.deepcopy(R.config_carlike_min_time(n))
    ocfg.model = 2
    s = m.BatchSolver(m.config_carlike_min_time(n, model=A.MODEL_SIMPLE_CAR_FRONT), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    _summary(r, ref, B)
    match, other = account(f"front-wheel car, B={B}", ocfg, (x0, xf, up, dtp), r, ref)
    assert match.sum() >= 0.8 * (r.status == 0).sum()
    s.close()


def test_host_pointer_entry_costs_little_more_than_the_kernel(m):
    """mpc_solve_batch (host pointers) stages through ONE pinned block each way: at B = 4096 the call must not take more than 1.15 x the
    solve kernel it wraps (it was 2.5 x with seven pageable copies in, five out)."""
    B, n = 4096, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    s.solve(x0, xf, up, dtp)
    walls, kern = [], []
    for _ in range(5):
        t = time.perf_counter()
        r = s.solve(x0, xf, up, dtp)
        walls.append(time.perf_counter() - t); kern.append(s.last_kernel_ms() * 1e-3)
    ratio = min(walls) / np.median(kern)
    print(f"[host entry] B={B}: wall {min(walls) * 1e3:.2f} ms, kernel {np.median(kern) * 1e3:.2f} ms, ratio {ratio:.3f}")
    assert ratio <= 1.15
    s.close()


def polygon_obstacles(x0, xf, seed, n_obst=4, V=5, lo=0.45, hi=1.1, lines=False):
    """convex polygons (or, with lines=True, alternating 2-vertex line obstacles and polygons) beside the straight start-goal line"""
    rng = np.random.default_rng(seed)
    B = x0.shape[0]
    d = xf[:, :2] - x0[:, :2]
    nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    nv = np.zeros((B, n_obst), np.int32); vt = np.zeros((B, n_obst, V, 2))
    for b in range(B):
        for o in range(n_obst):
            rad = rng.uniform(0.12, 0.3)
            c = x0[b, :2] + rng.uniform(0.2, 0.8) * d[b] + rng.choice([-1.0, 1.0]) * rng.uniform(rad + lo, rad + hi) * nrm[b]
            k = 2 if (lines and o % 2 == 0) else int(rng.integers(3, V + 1))
            ang = np.sort(rng.uniform(0, 2 * np.pi, k)) if k > 2 else rng.uniform(0, np.pi) + np.array([0.0, np.pi])
            nv[b, o] = k
            vt[b, o, :k, 0] = c[0] + rad * np.cos(ang); vt[b, o, :k, 1] = c[1] + rad * np.sin(ang)
    return np.full(B, n_obst, np.int32), nv, vt


@pytest.mark.parametrize("name,lines", [("line", False), ("line", True), ("polygon", False)])
def test_line_and_polygon_footprints_against_line_and_polygon_obstacles(m, c_oracle, name, lines):
    """a21, the pairing VERDICT r1 found missing: the car-like example's line footprint (cfg/carlike/mpc_local_planner_params.yaml:19-23) and
    the polygon footprint against costmap_converter-style polygon / line obstacles (src/mpc_local_planner_ros.cpp:501-541) --
    teb distance_segment_to_polygon_2d / distance_polygon_to_polygon_2d -- car-like n = 50, against the C oracle + KKT accounting."""
    from oracle import se2_nlp as R
    B, n, O, V = (128 if name == "line" else 64), 50, 4, 5
    kind, params, dmin = FOOTPRINTS[name]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=921, goal_range=(2.0, 5.0))
    no, nv, vt = polygon_obstacles(x0, xf, 922, O, V, lines=lines)
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = kind, params, dmin, 0.5, 2.5
    kw = dict(footprint_kind=kind, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=O, max_vertices=V, max_obstacle_rows=4)
    kw.update(dict(footprint_vertices=params) if kind == 4 else dict(footprint_params=params))
    s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt), obst=c_oracle.obst_from_nlp_config(ocfg, O, V, 4))
    both = _summary(r, ref, B)
    # the obstacles shape a good share of the solutions (otherwise this test would say nothing about the rows)
    free = c_oracle.solve_batch(c_oracle.from_nlp_config(R.config_carlike_min_time(n)), x0, xf, up, dtp)
    moved = both & (free[3] == 0) & (np.abs(ref[0] - free[0]).reshape(B, -1).max(1) > 1e-3)
    assert moved.sum() >= 0.15 * B
    account(f"{name} footprint vs {'line+polygon' if lines else 'polygon'} obstacles, B={B}", ocfg, (x0, xf, up, dtp), r, ref, obstacles=(no, nv, vt), max_rows=4)
    s.close()


@pytest.mark.parametrize("name", ["line", "two_circles"])
def test_dynamic_obstacles_with_turning_footprints(m, c_oracle, name):
    """a22 with a footprint that turns with the pose (stage_inequality_se2.cpp:177-189 through Line / TwoCirclesRobotFootprint): rows carry heading
    AND dt parts (x-dt, y-dt, theta-dt, dt-dt)."""
    from oracle import se2_nlp as R
    B, n = 128, 50
    kind, params, dmin = FOOTPRINTS[name]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=931, goal_range=(2.0, 5.0))
    no, nv, vt = point_obstacles(x0, xf, 932, n_obst=3, lo=0.6, hi=1.1)
    rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
    d = xf[:, :2] - x0[:, :2]
    nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.2 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params = kind, params
    ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, dmin, 0.5, 2.5
    s = m.BatchSolver(m.config_carlike_min_time(n, footprint_kind=kind, footprint_params=params, enable_dynamic_obstacles=True, min_obstacle_dist=dmin,
                                                force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel))
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel), obst=c_oracle.obst_from_nlp_config(ocfg, 3, 1, 4))
    both = _summary(r, ref, B, min_frac=0.45)      # a hard workload for the interior-point method from the reference guess alone (~60 % converge within 100 iterations, in every implementation)
    still = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt, rad, 0 * vel), obst=c_oracle.obst_from_nlp_config(ocfg, 3, 1, 4))
    assert (both & (still[3] == 0) & (np.abs(ref[0] - still[0]).reshape(B, -1).max(1) > 1e-4)).sum() >= 5      # the motion matters
    account(f"dynamic obstacles, {name} footprint, B={B}", ocfg, (x0, xf, up, dtp), r, ref, obstacles=(no, nv, vt, rad, vel), max_rows=4)
    s.close()
    # r03 (VERDICT r02 item 1 iii): the reference path alone converges for ~60 % of these instances (45 % before the clearance rows' slack start of 0.5);
    # hedged by three Hermite seeds -- candidate initial trajectories, BASELINE north star -- for >= 95 %; the hedges' answers are KKT points of the same NLP
    sc = m.BatchSolver(m.config_carlike_min_time(n, footprint_kind=kind, footprint_params=params, enable_dynamic_obstacles=True, min_obstacle_dist=dmin,
                                                 force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4,
                                                 candidates=(0, 5, 5, 6), candidate_max_iter=(100, 100, 100, 100), candidate_param=(0.0, 2.0, 1.0, 2.0)), max_batch=B)
    rc = sc.solve(x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel))
    wc, _ = sc.last_candidates(B)
    sc.close()
    print(f"dynamic obstacles, {name} footprint: reference path alone {np.mean(r.status == 0):.3f} converged, with hedges {np.mean(rc.status == 0):.3f}; winners {np.bincount(wc + 1, minlength=5).tolist()}")
    assert np.mean(r.status == 0) >= 0.5 and np.mean(rc.status == 0) >= 0.95
    same0 = (wc == 0) & (r.status == 0)
    assert (wc[r.status == 0] == 0).all() and np.array_equal(rc.x[same0], r.x[same0])          # the reference path's answer wherever it has one
    # the clearance rows of a solve are associated on the trajectory it STARTS from (StageInequalitySE2::update runs in the grid update): a hedge's answer is a KKT point of the
    # NLP with the rows of ITS seed -- the checker is told which trajectory every answer started from (oracle/candidates.py restates the seeds)
    from oracle import candidates as OC
    seeds = {c: OC.guess(k, x0, xf, n, ocfg.dt_ref, param=p)[0] for c, (k, p) in enumerate(((5, 2.0), (5, 1.0), (6, 2.0)), start=1)}
    start_x = np.stack([seeds[int(wc[i])][i] if wc[i] > 0 else R.cold_start(ocfg, x0[i], xf[i]).x for i in range(B)])
    # accounting in two parts: the instances candidate 0 answers are the reference path's answers -- the usual floor on the share that matches the
    # C oracle applies to them; a hedge's answer has no counterpart on the oracle's single path, so for those only the classification counts (KKT point of the NLP with its rows)
    sub = lambda a, i: None if a is None else a[i]
    for part, idx, floor in (("candidate 0", np.nonzero(wc == 0)[0], 0.7), ("hedges", np.nonzero(wc != 0)[0], 0.0)):
        if len(idx) == 0:
            continue
        rr = m.BatchResult(rc.x[idx], rc.u[idx], rc.dt[idx], rc.status[idx], rc.iters[idx])
        account(f"dynamic obstacles, {name} footprint, with hedges: answered by {part}, B={len(idx)}", ocfg, (x0[idx], xf[idx], up[idx], dtp[idx]), rr, tuple(a[idx] for a in ref[:5]),
                obstacles=tuple(sub(a, idx) for a in (no, nv, vt, rad, vel)), max_rows=4, min_match=floor, start_x=start_x[idx])


def test_rows_that_do_not_fit_are_counted(m, c_oracle):
    """More obstacles inside force_inclusion_dist than max_obstacle_rows (a dense costmap): the solver keeps the closest ones and reports,
    per instance, how many rows did not fit -- the same count as the C oracle's restatement of the rule; with enough rows the count is 0."""
    from oracle import se2_nlp as R
    B, n, O = 64, 30, 24
    x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(B, seed=941, goal_range=(1.5, 2.5))
    rng = np.random.default_rng(942)
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.1, 0.9, (B, O, 1)) * d + rng.uniform(0.3, 0.55, (B, O, 1)) * rng.choice([-1.0, 1.0], (B, O, 1)) * nrm
    no = np.full(B, O, np.int32); nv = np.ones((B, O), np.int32); vt = pts.reshape(B, O, 1, 2)
    ocfg = R.config_unicycle_quadratic(n)
    ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = 0.2, 0.5, 2.5
    for M, expect_drop in ((4, True), (16, False)):
        s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=1, max_obstacle_rows=M), max_batch=B)
        r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
        dev = s.last_rows_dropped(B)
        ora = np.zeros(B, np.int32)
        ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, vt), obst=c_oracle.obst_from_nlp_config(ocfg, O, 1, M), rows_dropped=ora)
        np.testing.assert_array_equal(dev, ora)
        assert (dev > 0).any() == expect_drop
        both = (r.status == 0) & (ref[3] == 0)
        assert both.sum() >= 0.5 * B and np.median(np.abs(r.x - ref[0]).reshape(B, -1).max(1)[both]) < 1e-7
        if not expect_drop:       # every wanted row is there: clearance holds against ALL obstacles
            xs = r.x[both][:, 1:-1, None, :2]
            dist = np.linalg.norm(xs - pts[both][:, None, :, :], axis=-1)
            assert dist.min() > 0.2 - 1e-6
        s.close()


def test_feasibility_check_bit_exact_vs_numpy_restatement(m):
    """8(f)-2: mpc_check_feasibility (Controller::isPoseTrajectoryFeasible on the device) against oracle/feasibility.py, flag for flag, on
    random costmaps (lethal + no-information cells), solver trajectories that partly leave the map, polygon / 2-point footprints,
    restricted look-ahead and per-instance grid sizes."""
    from oracle import feasibility as FO
    B, n, sx, sy, res = 96, 30, 90, 70, 0.1
    rng = np.random.default_rng(951)
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=952, goal_range=(1.5, 4.0))
    s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    cost = (rng.random((B, sy, sx)) < 0.004).astype(np.uint8) * 254
    cost[rng.random((B, sy, sx)) < 0.002] = 255
    cost[rng.random((B, sy, sx)) < 0.01] = 253
    origin = np.stack([rng.uniform(-5.0, -3.5, B), rng.uniform(-4.0, -3.0, B)], 1)          # some trajectories leave the 9 m x 7 m window
    spec = np.array([(0.25, 0.15), (-0.2, 0.15), (-0.2, -0.15), (0.25, -0.15), (0.32, 0.0)])
    for sp, la, ngrid in ((spec, -1, None), (spec, 9, None), (spec[:2], -1, None), (spec, -1, rng.integers(3, n + 1, B).astype(np.int32))):
        s.set_grid_sizes(ngrid)
        dev = s.check_feasibility(r.x, cost, res, origin, sp, 0.15, 0.3, la)
        ref = np.array([FO.is_pose_trajectory_feasible(cost[b], res, origin[b], r.x[b][: (n if ngrid is None else ngrid[b])], sp, 0.15, 0.3, la) for b in range(B)], np.int32)
        np.testing.assert_array_equal(dev, ref)
        assert 0 < ref.sum() < B          # both outcomes occur
    s.set_grid_sizes(None)
    s.close()


def test_convexified_hessian_reference_like_mode(m, c_oracle):
    """MPC_HESSIAN_CONVEXIFIED (+ tol 1e-4 = the "reference-like" setting: the car-like example YAML runs Ipopt with tol 1e-4 and a positive
    definite limited-memory Hessian, cfg/carlike/mpc_local_planner_params.yaml:91-95): every stage block of the constraint curvature is
    replaced by its positive semidefinite part.  Device vs the C oracle running the same mode (config 2, B = 256): same converged set, same
    iterates; and -- what the mode is for -- (almost) no regularisation retries, at the price of more iterations than the exact Hessian."""
    from oracle import se2_nlp as R
    from mpc_local_planner_amd import _abi as A
    B, n = 256, 50
    inputs = m.workloads.carlike_min_time_inputs(B)
    ocfg = R.config_carlike_min_time(n)
    for tol in (1e-8, 1e-4):
        s = m.BatchSolver(m.config_carlike_min_time(n, hessian_mode=1, tol=tol), max_batch=B)
        r = s.solve(*inputs)
        ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, tol=tol, hessian_mode=1), *inputs)
        both = (r.status == 0) & (ref[3] == 0)
        err = np.abs(r.x - ref[0]).reshape(B, -1).max(1)
        print(f"[convexified Hessian, tol {tol:g}] converged device {np.mean(r.status == 0):.3f} oracle {np.mean(ref[3] == 0):.3f}; iterations device {r.iters.mean():.1f} oracle {ref[4].mean():.1f}; "
              f"median |d| {np.median(err[both]):.1e}; within 1e-4: {np.mean(err[both] < 1e-4):.3f}")
        assert (r.status == ref[3]).mean() > 0.93 and np.median(err[both]) < (1e-7 if tol < 1e-6 else 1e-5)
        if tol < 1e-6:
            account(f"convexified Hessian, B={B}", ocfg, inputs, r, ref)
        s.close()
    se = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    re = se.solve(*inputs)
    assert r.iters.mean() < 1.6 * re.iters.mean()
    se.close()


def test_candidates_compose_with_clearance_rows(m, c_oracle):
    """Candidate initial trajectories on a problem WITH obstacles (line footprint, polygons): every candidate associates its own clearance rows on
    its own initial trajectory; the result is deterministic, candidate 0 == the single-candidate solve wherever that converges within its cap,
    the converged fraction does not fall, and every converged result keeps the clearance in the reference-form distance."""
    from oracle import se2_nlp as R, kkt_check as KC
    from mpc_local_planner_amd import _abi as A
    B, n, O, V = 128, 50, 4, 5
    kind, params, dmin = FOOTPRINTS["line"]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=921, goal_range=(2.0, 5.0))
    no, nv, vt = polygon_obstacles(x0, xf, 922, O, V)
    kw = dict(footprint_kind=kind, footprint_params=params, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=O, max_vertices=V, max_obstacle_rows=4)
    s1 = m.BatchSolver(m.config_carlike_min_time(n, max_iter=60, **kw), max_batch=B)
    r1 = s1.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    s1.close()
    s = m.BatchSolver(m.config_carlike_min_time(n, candidates=(A.CAND_REFERENCE, A.CAND_HERMITE_FF, A.CAND_BLEND), candidate_max_iter=(60, 60, 60), candidate_param=(0, 2.0, 0), **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    win, tot = s.last_candidates(B)
    r2 = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    np.testing.assert_array_equal(r.x, r2.x)
    ok1 = r1.status == 0
    assert (win[ok1] == 0).all() and np.array_equal(r.x[ok1], r1.x[ok1])
    print(f"[candidates + clearance rows] converged: single {ok1.mean():.3f}, with candidates {np.mean(r.status == 0):.3f}; winners {np.bincount(win + 1, minlength=4).tolist()}")
    assert np.mean(r.status == 0) >= ok1.mean() + 0.05
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params = kind, params
    for b in np.nonzero(r.status == 0)[0][:48]:
        obs = KC.obstacle_list(no[b], nv[b], vt[b])
        d = min(R.footprint_distance(kind, params, r.x[b, k], o) for k in range(1, n - 1) for o in obs)
        # rows exist only for the ASSOCIATED obstacles (nearest left / right + forced ones of the start trajectory): allow what the reference allows
        assert d > -1e-9
    s.close()


def test_abi_guards_of_per_instance_state(m):
    """solves must not read per-instance state (grid sizes) that was set for a smaller batch; a partial initial-guess triple is an
    error, not a silent cold start; the costmap kernel does not disturb the solve kernel's timer."""
    import ctypes as C
    from mpc_local_planner_amd._abi import MPC_EBATCH, MPC_EINVAL
    B, n = 8, 20
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=3, goal_range=(1.0, 2.0))
    s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    s.set_grid_sizes(np.full(4, 12, np.int32))
    with pytest.raises(m.MpcError) as e:
        s.solve(x0, xf, up, dtp)
    assert e.value.code == MPC_EBATCH
    r = s.solve(x0[:4], xf[:4], up[:4], dtp[:4])
    assert r.x.shape == (4, n, 3)
    s.set_grid_sizes(None)
    r = s.solve(x0, xf, up, dtp)
    xo = np.empty((B, n, 3)); uo = np.empty((B, n, 2)); do = np.empty(B)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = s._lib.mpc_solve_batch(s._h, B, p(x0), p(xf), p(up), p(dtp), p(r.x), None, None, None, p(xo), p(uo), p(do), None, None)
    assert rc == MPC_EINVAL and b"all three or none" in s._lib.mpc_last_error()
    s.close()
    so = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=8, max_vertices=1), max_batch=B)
    xq = m.workloads.unicycle_quadratic_inputs(B, seed=4)
    no = np.zeros(B, np.int32); nv = np.zeros((B, 8), np.int32); vt = np.zeros((B, 8, 1, 2))
    so.solve(*xq, obstacles=(no, nv, vt))
    t_solve = so.last_kernel_ms()
    so.costmap_to_obstacles(np.zeros((B, 30, 30), np.uint8), 0.1, np.zeros((B, 2)), np.zeros((B, 3)))
    assert so.last_kernel_ms() == t_solve and t_solve > 0
    so.close()


def test_candidate_results_do_not_depend_on_batch_composition_or_timing(m):
    """Hedging is timing dependent (which hedges start, how far they get), the RESULT must not be: instances solved inside a batch of 4096 (every
    candidate-0 workgroup is dispatched before any hedge) and the same instances solved as batches of 1024 and of 64 (hedges start at once)
    give bit-identical trajectories, statuses, iteration counts and winners."""
    from mpc_local_planner_amd import _abi as A
    n, B = 50, 4096
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=955)
    s = m.BatchSolver(m.config_carlike_min_time(n, candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)), max_batch=B)
    big = s.solve(x0, xf, up, dtp)
    wbig, _ = s.last_candidates(B)
    for lo, cnt in ((0, 1024), (3000, 64), (4095, 1)):
        sl = slice(lo, lo + cnt)
        r = s.solve(x0[sl], xf[sl], up[sl], dtp[sl])
        w, _ = s.last_candidates(cnt)
        np.testing.assert_array_equal(r.x, big.x[sl]); np.testing.assert_array_equal(r.u, big.u[sl]); np.testing.assert_array_equal(r.dt, big.dt[sl])
        np.testing.assert_array_equal(r.status, big.status[sl]); np.testing.assert_array_equal(r.iters, big.iters[sl]); np.testing.assert_array_equal(w, wbig[sl])
    assert (big.status == 0).mean() > 0.99
    s.close()


DEVICE_COST_VARIANTS = {
    # name -> make_config keywords; the oracle-side configuration of the same name is tests/test_oracle_solver.py::cost_variant
    "full_weights": dict(Q=[[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.4]], R=[[0.1, 0.02], [0.02, 0.05]], Qf=[[8.0, 1.0, 0.0], [1.0, 9.0, 0.5], [0.0, 0.5, 0.6]]),
    "trapezoid_fixed_dt": dict(integral_form=True, cost_integration=1),
    "trapezoid_free_dt": dict(integral_form=True, cost_integration=1, dt_free=True, dt_lb=0.05, dt_ub=1.0),
    "trapezoid_xf_fixed_free_dt": dict(integral_form=True, cost_integration=1, xf_fixed=(True, True, True), dt_free=True, dt_lb=0.05, dt_ub=1.0),
    "hybrid": dict(Q=(0, 0, 0), Qf=None, hybrid_cost_minimum_time=True, dt_free=True, xf_fixed=(True, True, True), R=(1.0, 0.5)),
    "hybrid_integral": dict(Q=(0, 0, 0), Qf=None, hybrid_cost_minimum_time=True, dt_free=True, xf_fixed=(True, True, True), R=(1.0, 0.5), integral_form=True),
    "all": dict(Q=[[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.4]], R=[[0.1, 0.02], [0.02, 0.05]], Qf=[[8.0, 1.0, 0.0], [1.0, 9.0, 0.5], [0.0, 0.5, 0.6]], integral_form=True,
                cost_integration=1, terminal_ball_S=[[1.0, 0.2, 0.0], [0.2, 1.0, 0.1], [0.0, 0.1, 0.5]], terminal_ball_gamma=0.3, dt_free=True, dt_lb=0.05, dt_ub=1.0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DEVICE_COST_VARIANTS))
def test_cost_variants_vs_c_oracle(m, c_oracle, name):
    """full Q / R / Qf / S weight matrices (src/controller.cpp:561-592,652-668,686-702), trapezoidal rule for integral-form costs
    (finite_differences_grid_se2.cpp:63-68), hybrid minimum time + control cost (src/controller.cpp:616-618): device against the C oracle on
    B = 128 instances of the unicycle quadratic-form workload, accounting (match / other KKT point of the reference-form NLP / unclassified = 0)."""
    from test_oracle_solver import cost_variant
    B, n = 128, 16
    ocfg = cost_variant(name, n)
    inputs = m.workloads.unicycle_quadratic_inputs(B, seed=11)
    s = m.BatchSolver(m.config_unicycle_quadratic(n, **DEVICE_COST_VARIANTS[name]), max_batch=B)
    r = s.solve(*inputs)
    out = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *inputs)
    match, other = account(f"cost variant {name}", ocfg, inputs, r, out)
    assert (r.status == 0).mean() > 0.95 and match.sum() > 0.9 * B
    assert abs(float(r.iters[match].mean()) - float(out[4][match].mean())) < 0.5          # same iterate sequences
    s.close()


def test_solver_from_a_parameter_file_of_the_reference(m, tmp_path):
    """BatchSolver.from_yaml: a parameter file in the reference's layout (tests/test_params.py::CARLIKE_YAML: car-like model, line footprint, Crank-Nicolson
    collocation, terminal cost + terminal ball with a free heading, convexified Hessian, tol 1e-4) -> a configured handle; the batch solves and the
    handle is sized for the largest grid the reference's grid adaptation may reach."""
    from test_params import CARLIKE_YAML
    f = tmp_path / "params.yaml"
    f.write_text(CARLIKE_YAML)
    B = 64
    s = m.BatchSolver.from_yaml(str(f), max_batch=B, max_obstacles=4, max_vertices=1)
    assert s.n == 60 and s.n_ref == 24 and s.controller_options["max_grid_size"] == 60 and any("limited-memory" in t for t in s.param_notes)
    assert (s.grid_sizes(B) == 24).all()
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=3, goal_range=(1.0, 3.0))
    no, nv, vt = point_obstacles(x0, xf, seed=9)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, vt))
    ok = r.status == 0
    assert ok.mean() > 0.7
    # the two goal components that the file fixes are met, the free heading stays inside the terminal ball (S = diag(1, 1, 0.5), radius 0.2)
    last = s.n_ref - 1
    assert np.abs(r.x[ok, last, :2] - xf[ok, :2]).max() < 1e-9
    dth = np.arctan2(np.sin(r.x[ok, last, 2] - xf[ok, 2]), np.cos(r.x[ok, last, 2] - xf[ok, 2]))
    assert (0.5 * dth ** 2 <= 0.2 + 1e-6).all()
    s.close()


@pytest.mark.parametrize("case", ["line_footprint_moving_obstacle_n80", "polygon_footprint_n64", "full_weight_matrices_n70", "terminal_ball_n70", "via_points_n60"])
def test_extended_levels_in_the_global_form_equal_the_lds_form_bit_for_bit(m, case):
    """(r06) mpc_config.stage_data for the EXTENDED kernel levels: turning footprints, moving obstacles, terminal ball, via-points and the cost variants exist
    in the global form of the factorisation data too (stage_inequality_se2.cpp:164-189 with a line / polygon footprint at n >= 60 used to run two workgroups per CU).  Same
    arithmetic on the same numbers: trajectories, controls, dt, statuses and iteration counts bit for bit, and MPC_STAGE_AUTO takes the global form where it puts more workgroups
    on a CU."""
    from mpc_local_planner_amd import _abi as A
    obstacles, via = None, None
    if case == "line_footprint_moving_obstacle_n80":
        B, n = 128, 80
        kind, params, dmin = FOOTPRINTS["line"]
        x0, xf, up, dtp, obstacles = m.workloads.carlike_moving_obstacle_inputs(B)          # (the workload of bench.py's leg carlike_n80_line_footprint_moving_obstacle)
        mk = lambda **k: m.config_carlike_min_time(n, footprint_kind=kind, footprint_params=params, enable_dynamic_obstacles=True, min_obstacle_dist=dmin, force_inclusion_dist=0.5,
                                                   cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4, **k)
    elif case == "polygon_footprint_n64":
        B, n = 128, 64
        kind, params, dmin = FOOTPRINTS["polygon"]
        x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=953, goal_range=(2.0, 6.0))
        obstacles = point_obstacles(x0, xf, 954)
        mk = lambda **k: m.config_carlike_min_time(n, footprint_kind=kind, footprint_vertices=params, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=4,
                                                   max_vertices=1, max_obstacle_rows=4, **k)
    elif case == "full_weight_matrices_n70":      # the "full_weights" variant of test_cost_variants_vs_c_oracle (level 2: serial sweeps only), on a grid of 70 points
        B, n = 128, 70
        x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(B, seed=11)
        mk = lambda **k: m.config_unicycle_quadratic(n, **DEVICE_COST_VARIANTS["full_weights"], **k)
    elif case == "terminal_ball_n70":             # the workload of test_terminal_ball_batch_vs_c_oracle on 70 points (level 1)
        B, n = 128, 70
        x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(B, seed=904, goal_range=(0.8, 1.3))
        mk = lambda **k: m.config_unicycle_quadratic(n, Q=(0.2, 0.2, 0.02), R=(1.0, 0.5), Qf=None, terminal_ball_S=(1.0, 1.0, 0.01), terminal_ball_gamma=0.02, **k)
    else:
        B, n = 128, 60
        x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=957, goal_range=(2.0, 6.0))
        rng = np.random.default_rng(958)
        VP = 3
        t = np.sort(rng.uniform(0.2, 0.8, (B, VP)), axis=1)
        vp = np.zeros((B, VP, 3))
        vp[..., :2] = x0[:, None, :2] + t[..., None] * (xf[:, None, :2] - x0[:, None, :2]) + rng.normal(0, 0.15, (B, VP, 2))
        via = (np.full(B, VP, np.int32), vp)
        mk = lambda **k: m.config_carlike_min_time(n, objective=m.OBJ_MIN_TIME_VIA_POINTS, vp_position_weight=0.5, max_via_points=VP, **k)
    res, lds = {}, {}
    for mode in (A.STAGE_LDS, A.STAGE_GLOBAL, A.STAGE_AUTO):
        s = m.BatchSolver(mk(stage_data=mode), max_batch=B)
        if via is not None:
            s.set_via_points(*via)
        res[mode] = s.solve(x0, xf, up, dtp, obstacles=obstacles)
        lds[mode] = s.lds_bytes()
        s.close()
    a, g, auto = res[A.STAGE_LDS], res[A.STAGE_GLOBAL], res[A.STAGE_AUTO]
    assert (a.status == 0).mean() > 0.5, (a.status == 0).mean()
    for f in ("status", "iters", "dt", "x", "u"):
        assert np.array_equal(getattr(a, f), getattr(g, f), equal_nan=True), (case, f)
        assert np.array_equal(getattr(a, f), getattr(auto, f), equal_nan=True), (case, f)
    per_cu = lambda b: min(4, (160 * 1024) // b)
    assert lds[A.STAGE_GLOBAL] < lds[A.STAGE_LDS]
    print(f"[extended levels, {case}] LDS form {lds[A.STAGE_LDS]} B = {per_cu(lds[A.STAGE_LDS])} per CU, global form {lds[A.STAGE_GLOBAL]} B = {per_cu(lds[A.STAGE_GLOBAL])} per CU, AUTO takes {lds[A.STAGE_AUTO]} B; converged {(a.status == 0).mean():.3f}")
    if per_cu(lds[A.STAGE_LDS]) <= 2 and per_cu(lds[A.STAGE_GLOBAL]) > per_cu(lds[A.STAGE_LDS]):
        assert lds[A.STAGE_AUTO] == lds[A.STAGE_GLOBAL]
