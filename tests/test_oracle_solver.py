"""CPU tests of the oracle's SOLVE: numpy dense IPM vs scipy on the reference-form NLP, the C oracle
(banded LU) vs the numpy IPM, and both against the committed golden fixtures."""
import os

import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import se2_nlp as R
from oracle import ipm_dense as I
from oracle import kkt_check as K

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OCFG = {
    "carlike_min_time_n50": lambda: R.config_carlike_min_time(50),
    "carlike_min_time_n20": lambda: R.config_carlike_min_time(20),
    "unicycle_quadratic_n20": lambda: R.config_unicycle_quadratic(20),
    "bicycle_min_time_n30": lambda: R.config_bicycle_min_time(30),
}


def _ipm(cfg, inp, **kw):
    return I.solve(cfg, inp, R.cold_start(cfg, inp.x0, inp.xf), opt=I.IpmOptions(max_iter=100, **kw))


def _slsqp_polish(cfg, inp, traj):
    """Independent check: SLSQP on the REFERENCE-form NLP (division by dt, reference row order)
    started at the IPM solution must stay there (it is a KKT point of the reference's NLP)."""
    nlp = R.ReferenceNlp(cfg, inp)
    z0 = nlp.pack(traj)
    lb, ub = nlp.bounds()
    bnds = [(None if l < -1e29 else l, None if u > 1e29 else u) for l, u in zip(lb, ub)]
    if cfg.dt_free:
        bnds[-1] = (1e-3, cfg.dt_ub)
    cons = [{"type": "eq", "fun": nlp.equalities}, {"type": "ineq", "fun": lambda z: -nlp.inequalities(z)}]
    r = minimize(nlp.objective, z0, method="SLSQP", bounds=bnds, constraints=cons, options=dict(maxiter=50, ftol=1e-12))
    return nlp, z0, r


def test_config1_unicycle_quadratic_single_instance_matches_scipy():
    # BASELINE.json config 1: the reference's own CPU-runnable case
    cfg = R.config_unicycle_quadratic(20)
    inp = R.CycleInputs(x0=np.array([0.0, 0.0, 0.0]), xf=np.array([1.0, 0.3, 0.2]), u_prev=np.zeros(2), dt_prev=0.2)
    res = _ipm(cfg, inp)
    assert res.status == 0 and res.kkt_error < 1e-8
    nlp, z0, r = _slsqp_polish(cfg, inp, res.traj)
    assert np.abs(nlp.equalities(z0)).max() < 1e-8
    assert nlp.inequalities(z0).max() < 1e-8
    assert np.abs(r.x - z0).max() < 1e-5
    assert abs(r.fun - res.objective) < 1e-6 * abs(res.objective)      # barrier residue mu=tol/10
    # also from the cold start scipy must find the same (convex-like) optimum
    zc = nlp.pack(R.cold_start(cfg, inp.x0, inp.xf))
    lb, ub = nlp.bounds()
    bnds = [(None if l < -1e29 else l, None if u > 1e29 else u) for l, u in zip(lb, ub)]
    cons = [{"type": "eq", "fun": nlp.equalities}, {"type": "ineq", "fun": lambda z: -nlp.inequalities(z)}]
    rc = minimize(nlp.objective, zc, method="SLSQP", bounds=bnds, constraints=cons, options=dict(maxiter=300, ftol=1e-13))
    assert abs(rc.fun - res.objective) < 1e-6 * max(1.0, abs(res.objective))
    assert np.abs(rc.x - z0).max() < 1e-4


def test_carlike_min_time_solution_is_kkt_point_of_reference_form():
    cfg = R.config_carlike_min_time(20)
    inp = R.CycleInputs(x0=np.array([0.0, 0.0, 0.2]), xf=np.array([1.5, 0.5, 0.4]), u_prev=np.array([0.1, 0.0]), dt_prev=0.2)
    res = _ipm(cfg, inp)
    assert res.status == 0
    nlp, z0, r = _slsqp_polish(cfg, inp, res.traj)
    assert np.abs(nlp.equalities(z0)).max() < 1e-7
    assert nlp.inequalities(z0).max() < 1e-7
    assert abs(r.fun - res.objective) < 1e-6        # SLSQP cannot improve it ...
    assert np.abs(r.x - z0).max() < 1e-3            # ... and stays put (its finite-difference gradients limit this to ~1e-4)
    # physically sensible: time optimal => some control saturates
    u = res.traj.u
    assert (np.abs(u[:, 0] - 0.4) < 1e-4).any()


def test_first_cycle_without_previous_control_drops_rate_rows_of_stage_zero():
    cfg = R.config_carlike_min_time(12)
    inp = R.CycleInputs(x0=np.array([0.0, 0.0, 0.0]), xf=np.array([1.0, 0.2, 0.1]), u_prev=np.zeros(2), dt_prev=0.0)
    res = _ipm(cfg, inp)
    assert res.status == 0
    # with dt_prev = 0 the first control may jump: it is at the speed limit immediately
    assert res.traj.u[0, 0] == pytest.approx(0.4, abs=1e-5)


@pytest.mark.parametrize("name", sorted(OCFG))
def test_numpy_ipm_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = OCFG[name]()
    for i in range(min(3, g["x0"].shape[0])):
        inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]))
        res = _ipm(cfg, inp)
        assert res.status == 0
        assert np.abs(res.traj.x - g["x"][i]).max() < 1e-6
        assert np.abs(res.traj.u - g["u"][i, :-1]).max() < 1e-6
        assert abs(res.traj.dt - g["dt"][i]) < 1e-8


def test_every_solver_fixture_records_how_it_was_made():
    """ every tests/golden/*.npz carries a `generator` entry: the script, the function, and for the interior-point fixtures the full option set
    (oracle/ipm_dense.py::IpmOptions) it was solved with"""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(GOLD, "*.npz")))
    assert len(files) >= 30
    for path in files:
        g = np.load(path)
        assert "generator" in g.files, path
        rec = json.loads(str(g["generator"]))
        assert rec["script"].startswith("tests/golden/make_") and os.path.exists(os.path.join(os.path.dirname(GOLD), "..", rec["script"])), (path, rec.get("script"))
        if rec["script"].endswith("make_golden.py"):
            assert rec["ipm_options"]["mu_strategy"] in ("adaptive", "monotone") and rec["ipm_options"]["tol"] == 1e-8, path


MERIT = {"carlike_min_time_n20_merit": lambda: R.config_carlike_min_time(20), "carlike_min_time_n50_merit": lambda: R.config_carlike_min_time(50),
         "bicycle_min_time_n30_merit": lambda: R.config_bicycle_min_time(30)}


@pytest.mark.parametrize("name", sorted(MERIT))
def test_answers_under_the_other_line_search(name, c_oracle):
    """tests/golden/*_merit.npz are made with the l1-MERIT line search (oracle_config.line_search = 0 / IpmOptions.globalization = "merit" / MPC_LS_MERIT: the globalisation of
    rounds 1-5) on the inputs of the base fixtures.  (1) Under the merit the C oracle and the numpy oracle reproduce them (same iterate sequence).  (2) The default solvers (Ipopt's
    filter line search) end, from the same start, at the same minimum on most instances -- same travel time to 1e-7 relative, states within 1e-4 --; where they do not, both
    answers are local minima of a multi-modal NLP (another travel time), which the test counts instead of hiding."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = MERIT[name]()
    B = g["x0"].shape[0]
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, line_search=0), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all() and np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6 and np.abs(do - g["dt"]).max() < 1e-8 and np.abs(it - g["iters"]).max() <= 1
    for i in range(2):
        inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]))
        res = I.solve(cfg, inp, R.cold_start(cfg, g["x0"][i], g["xf"][i]), opt=I.IpmOptions(max_iter=100, globalization="merit"))
        assert res.status == 0 and np.abs(res.traj.x - g["x"][i]).max() < 1e-6 and res.iters == g["iters"][i]
    xa, ua, da, sa, ia = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])      # the default: filter
    assert (sa == 0).all()
    same_T = np.abs(da - g["dt"]) < 1e-7 * np.abs(g["dt"])
    ex = np.abs(xa - g["x"]).reshape(B, -1).max(1)
    print(f"[{name}] filter vs merit answers: same travel time on {int(same_T.sum())} of {B}, max |x| difference there {ex[same_T].max():.1e}; iterations merit {g['iters'].tolist()} filter {ia.tolist()}")
    assert same_T.sum() >= B - 1 and ex[same_T].max() < 1e-4
    assert (ia != g["iters"]).any()              # the two line searches are different iterations


MONOTONE = {"carlike_min_time_n20_monotone": lambda: R.config_carlike_min_time(20), "unicycle_quadratic_n20_monotone": lambda: R.config_unicycle_quadratic(20)}


@pytest.mark.parametrize("name", sorted(MONOTONE))
def test_answers_under_the_other_barrier_rule(name, c_oracle):
    """(VERDICT r05 item 7 / ADVICE r04 item 4) One fixture set is made with mu_strategy = MONOTONE (Fiacco-McCormick, Ipopt's own default) instead of the adaptive rule every
    other fixture and every default solve uses.  (1) Under the monotone rule the numpy oracle and the C oracle reproduce it at 1e-6 (they follow the same iterate sequence:
    1e-12).  (2) The ADAPTIVE solvers, from the same start, end at the same minimum -- the objective agrees to 1e-7 relative -- but NOT at the same point to 1e-6 everywhere:
    minimum-time and effort-weighted optima have flat directions (controls sliding along weakly active rate rows), and two interior-point runs that stop at E_0 <= 1e-8 on
    different central paths sit up to 1e-4 apart in the controls, 2e-5 in the states (measured: 9 of 12 car-like and 3 of 8 unicycle instances within 1e-6, all states
    within 1e-4 = the tolerance BASELINE.json states).  That scatter is what "within 1e-4 of the Ipopt reference" can mean at best for these NLPs."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = MONOTONE[name]()
    B = g["x0"].shape[0]
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, mu_strategy=1), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all() and np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6 and np.abs(do - g["dt"]).max() < 1e-8 and np.abs(it - g["iters"]).max() <= 1
    for i in range(2):
        inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]))
        res = I.solve(cfg, inp, R.cold_start(cfg, g["x0"][i], g["xf"][i]), opt=I.IpmOptions(max_iter=100, mu_strategy="monotone"))
        assert res.status == 0 and np.abs(res.traj.x - g["x"][i]).max() < 1e-6 and np.abs(res.traj.u - g["u"][i, :-1]).max() < 1e-6
    xa, ua, da, sa, ia = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])      # the default: adaptive
    assert (sa == 0).all()
    ex = np.abs(xa - g["x"]).reshape(B, -1).max(1); eu = np.abs(ua - g["u"]).reshape(B, -1).max(1)
    assert ex.max() < 1e-4 and eu.max() < 3e-4, (ex.max(), eu.max())
    assert np.mean(np.maximum(ex, eu) < 1e-6) >= 0.3 and np.median(ex) < 2e-6
    if cfg.dt_free:
        assert np.abs(da - g["dt"]).max() < 1e-7 * np.abs(g["dt"]).max()          # same travel time
    print(f"[barrier rule, {name}] adaptive vs monotone answers: within 1e-6 {int(np.sum(np.maximum(ex, eu) < 1e-6))} of {B}, max state difference {ex.max():.1e}, max control difference {eu.max():.1e}")


@pytest.mark.parametrize("name", sorted(OCFG))
def test_c_oracle_reproduces_golden(name, c_oracle):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = OCFG[name]()
    oc = c_oracle.from_nlp_config(cfg)
    xo, uo, do, st, it = c_oracle.solve_batch(oc, g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6
    assert np.abs(uo - g["u"]).max() < 1e-6
    assert np.abs(do - g["dt"]).max() < 1e-8
    # banded-LU and dense solves follow the same iterate sequence
    assert np.abs(it - g["iters"]).max() <= 2


def test_c_oracle_warm_start_and_thread_invariance(c_oracle):
    from mpc_local_planner_amd import workloads as W
    cfg = R.config_carlike_min_time(20)
    oc = c_oracle.from_nlp_config(cfg)
    x0, xf, up, dtp = W.carlike_min_time_inputs(12, seed=5, goal_range=(1.0, 2.5))
    a = c_oracle.solve_batch(oc, x0, xf, up, dtp, nthreads=1)
    b = c_oracle.solve_batch(oc, x0, xf, up, dtp, nthreads=4)
    for p, q in zip(a, b):
        np.testing.assert_array_equal(p, q)
    ok = a[3] == 0
    # warm-start path (vertex values handed in, x_0 / fixed goal overwritten): every converged result is a
    # feasible trajectory of the same problem
    w = c_oracle.solve_batch(oc, x0, xf, up, dtp, init=(a[0], a[1], a[2]))
    okw = w[3] == 0
    assert okw.sum() >= 6
    for i in np.nonzero(okw)[0]:
        inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
        nlp = R.ReferenceNlp(cfg, inp)
        z = nlp.pack(R.Trajectory(w[0][i], w[1][i, :-1], float(w[2][i])))
        assert np.abs(nlp.equalities(z)).max() < 1e-6
        assert nlp.inequalities(z).max() < 1e-6
        np.testing.assert_array_equal(w[0][i, 0], x0[i])
        np.testing.assert_array_equal(w[0][i, -1], xf[i])


def test_numpy_ipm_with_obstacle_rows_reproduces_golden_and_keeps_clearance():
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_obstacles_n30.npz"))
    cfg = R.config_unicycle_quadratic(30)
    M = int(g["max_rows"])
    for i in range(2):
        nv = g["n_vertices"][i]
        obs = [R.Obstacle(R.OBST_POLYGON, g["vertices"][i, o, :nv[o]]) for o in range(g["n_obstacles"][i])]
        inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]), obstacles=obs)
        init = R.cold_start(cfg, inp.x0, inp.xf)
        rel, _ = R.associate_obstacles(cfg, init, obs, max_rows=M)
        assert max(len(q) for q in rel) <= M and rel[0] == []
        res = I.solve(cfg, inp, init, relevant=rel, opt=I.IpmOptions(max_iter=100))
        assert res.status == 0
        assert np.abs(res.traj.x - g["x"][i]).max() < 1e-6
        # reference-form rows: d_min - dist <= 0 for every associated obstacle
        nlp = R.ReferenceNlp(cfg, inp, relevant=rel)
        gz = nlp.inequalities(nlp.pack(res.traj))
        assert gz.max() < 1e-7


def test_dual_start_lands_on_the_same_solution():
    """Oracle-only feature for the next step of the warm start: multipliers of the previous cycle carried
    over as max(previous, mu0 / slack).  Second control cycle of a golden instance: same solution, not more iterations + 2."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "carlike_min_time_n20.npz"))
    cfg = R.config_carlike_min_time(20)
    i, per = 3, 0.2
    inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]))
    r1 = I.solve(cfg, inp, R.cold_start(cfg, inp.x0, inp.xf), opt=I.IpmOptions(max_iter=100))
    assert r1.status == 0 and r1.piL is not None
    u0 = r1.traj.u[0]
    x1 = inp.x0 + per * R.dynamics(cfg.model, cfg.model_params, inp.x0, u0)
    x1[2] = R.normalize_theta(x1[2])
    init = R.Trajectory(r1.traj.x.copy(), r1.traj.u.copy(), r1.traj.dt)
    init.x[0] = x1
    inp2 = R.CycleInputs(x0=x1, xf=g["xf"][i], u_prev=u0, dt_prev=per)
    opt = I.IpmOptions(max_iter=100, mu_init=1e-3)
    a = I.solve(cfg, inp2, init, opt=opt)
    b = I.solve(cfg, inp2, init, opt=opt, dual_start=r1)
    assert a.status == 0 and b.status == 0
    # the travel time is determined to the solver tolerance; the states of a minimum-time solution have flat directions (a KKT error of 1e-9 leaves ~1e-6 of play)
    assert np.abs(a.traj.x - b.traj.x).max() < 1e-5 and abs(a.traj.dt - b.traj.dt) < 1e-9
    assert b.iters <= a.iters + 2


def test_c_oracle_clearance_rows_match_numpy_goldens():
    """oracle/mpc_oracle.c restates the clearance rows independently of ipm_dense.py (teb distances, association, condensed rows in the
    banded KKT): it must land on the numpy oracle's golden solutions with the same iteration counts."""
    from oracle import c_oracle as CO
    here = os.path.dirname(os.path.abspath(__file__))
    for name, n in (("unicycle_quadratic_obstacles_n30", 30), ("unicycle_quadratic_obstacles_n80", 80)):
        g = np.load(os.path.join(here, "golden", name + ".npz"))
        cfg = R.config_unicycle_quadratic(n)
        O, V = g["vertices"].shape[1], g["vertices"].shape[2]
        xo, uo, do, st, it = CO.solve_batch(CO.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"],
                                            obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"]),
                                            obst=CO.obst_from_nlp_config(cfg, O, V, int(g["max_rows"])))
        assert (st == 0).all()
        assert np.abs(xo - g["x"]).max() < 1e-9
        assert (it == g["iters"]).all()


@pytest.mark.parametrize("name,kind,n,method", [
    ("carlike_min_time_midpoint_n20", "carlike", 20, 1), ("unicycle_quadratic_midpoint_n20", "unicycle", 20, 1),
    ("bicycle_min_time_midpoint_n30", "bicycle", 30, 1), ("carlike_min_time_cn_n20", "carlike", 20, 2),
    ("unicycle_quadratic_cn_n20", "unicycle", 20, 2), ("bicycle_min_time_cn_n30", "bicycle", 30, 2)])
def test_c_oracle_midpoint_and_crank_nicolson_match_numpy_goldens(name, kind, n, method, c_oracle):
    """oracle/mpc_oracle.c's stage_map (midpoint / literal crank-nicolson rows, chain-rule derivatives incl. the dt column and
    d2/ddt2) against the numpy oracle's fixtures (tests/golden/make_golden.py --midpoint / --cn)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {"carlike": R.config_carlike_min_time, "unicycle": R.config_unicycle_quadratic, "bicycle": R.config_bicycle_min_time}[kind](n)
    cfg.collocation = method
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6 and np.abs(do - g["dt"]).max() < 1e-8
    assert np.abs(it - g["iters"]).max() <= 2


def test_via_point_association_rules():
    """MinTimeViaPointsCost::update (min_time_via_points_cost.cpp:39-117) + findClosestPose (...grid_base_se2.cpp:364-388)."""
    x = np.zeros((6, 3))
    x[:, 0] = np.arange(6.0)                       # states on the x axis at 0..5 (the last one is the goal)
    cfg = R.config_carlike_min_time(6)
    cfg.objective = R.OBJ_MIN_TIME_VIA_POINTS
    assert R.find_closest_pose(x, 2.4, 1.0) == 2 and R.find_closest_pose(x, 2.5, 0.0) == 2       # first minimum wins a tie
    assert R.find_closest_pose(x, 9.0, 0.0) == 5 and R.find_closest_pose(x, 2.0, 0.0, start_idx=4) == 4
    vps = np.array([[2.4, 0.5, 0.0], [-1.0, 0.0, 0.0], [7.0, 0.0, 0.0], [0.9, 0.0, 0.0]])
    assert R.associate_via_points(cfg, x, vps) == [2, -1, 4, 1]                # behind the start: skipped; at/after the goal: n-2
    cfg.via_points_ordered = True
    # ordered: the search restarts two states behind the previous (unclamped) match; a match at the start moves to state 1
    assert R.associate_via_points(cfg, x, vps) == [2, 4, 4, 4]
    assert R.associate_via_points(cfg, x, np.array([[-1.0, 0, 0], [0.1, 0, 0], [3.2, 0, 0]])) == [1, 2, 4]


@pytest.mark.parametrize("name", ["carlike_via_points_n30", "carlike_via_points_ordered_n30", "unicycle_quadratic_ball_n20"])
def test_numpy_oracle_reproduces_via_point_and_terminal_ball_goldens(name):
    """regression pin of oracle/ipm_dense.py for the fixtures of tests/golden/make_golden.py --via / --ball (first two instances)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    for i in range(2):
        if "ball" in name:
            cfg = R.config_unicycle_quadratic(20)
            cfg.Q, cfg.R, cfg.Qf, cfg.terminal_ball_S, cfg.terminal_ball_gamma = g["Q"], g["R"], None, g["S"], float(g["gamma"])
            inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]))
        else:
            cfg = R.config_carlike_min_time(30)
            cfg.objective, cfg.vp_position_weight, cfg.vp_orientation_weight = R.OBJ_MIN_TIME_VIA_POINTS, float(g["wp"]), float(g["wo"])
            cfg.via_points_ordered = bool(g["ordered"])
            vps = g["via"][i, :int(g["n_via"][i])]
            inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]), via_points=vps)
            assert R.associate_via_points(cfg, R.cold_start(cfg, g["x0"][i], g["xf"][i]).x, vps) == list(g["idx"][i, :len(vps)])
        r = I.solve(cfg, inp, R.cold_start(cfg, g["x0"][i], g["xf"][i]), opt=I.IpmOptions(max_iter=100))
        assert r.status == 0 and r.iters == g["iters"][i]
        assert np.abs(r.traj.x - g["x"][i]).max() < 1e-9 and abs(r.traj.dt - g["dt"][i]) < 1e-10


def test_costmap_oracle_order_and_filter():
    """oracle/costmap.py against a hand-worked map (src/mpc_local_planner_ros.cpp:474-499): container order is x-index outer, y-index
    inner; the last row / column are never visited; cells behind the robot and farther than the limit are dropped."""
    from oracle import costmap as OC
    cost = np.zeros((4, 5), np.uint8)                 # size_y = 4, size_x = 5
    cost[2, 1] = 254; cost[0, 1] = 254; cost[1, 3] = 254; cost[0, 0] = 253; cost[3, 2] = 254; cost[1, 4] = 254
    pts = OC.costmap_to_obstacles(cost, 1.0, (10.0, 20.0), (12.0, 21.0, 0.0), behind_robot_dist=100.0)
    np.testing.assert_array_equal(pts, [[11.5, 20.5], [11.5, 22.5], [13.5, 21.5]])          # (1,0), (1,2), (3,1); row 3 and column 4 skipped
    # robot at x = 12 heading +x: the two cells at x = 11.5 are behind it; (1,2) is 1.58 away, (1,0) 0.71
    pts = OC.costmap_to_obstacles(cost, 1.0, (10.0, 20.0), (12.0, 21.0, 0.0), behind_robot_dist=1.0)
    np.testing.assert_array_equal(pts, [[11.5, 20.5], [13.5, 21.5]])
    assert OC.costmap_to_obstacles(np.full((1, 1), 254, np.uint8), 1.0, (0, 0), (0, 0, 0)).shape == (0, 2)


@pytest.mark.parametrize("name", ["carlike_via_points_n30", "carlike_via_points_ordered_n30"])
def test_c_oracle_via_points_match_numpy_goldens(name, c_oracle):
    """oracle/mpc_oracle.c restates the via-point association and objective terms independently of ipm_dense.py."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = R.config_carlike_min_time(30)
    cfg.objective, cfg.vp_position_weight, cfg.vp_orientation_weight = R.OBJ_MIN_TIME_VIA_POINTS, float(g["wp"]), float(g["wo"])
    cfg.via_points_ordered = bool(g["ordered"])
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"], via=(g["n_via"], g["via"]))
    assert (st == 0).all()
    # one instance of the ordered set ends in a flat valley of its minimum (same iteration count, dt equal to 3e-13, states 6e-5 apart between the two linear algebras): the travel
    # time pins dt, the states along the valley only to the KKT tolerance
    xtol = 1e-4 if name.endswith("ordered_n30") else 1e-6
    assert np.abs(xo - g["x"]).max() < xtol and np.abs(uo - g["u"]).max() < 10 * xtol and np.abs(do - g["dt"]).max() < 1e-8
    assert (np.abs(xo - g["x"]).reshape(len(st), -1).max(1) < 1e-6).sum() >= len(st) - 1
    assert np.abs(it - g["iters"]).max() <= 2


@pytest.mark.parametrize("name,free", [("unicycle_quadratic_integral_n20", False), ("unicycle_quadratic_integral_free_dt_n20", True)])
def test_c_oracle_integral_form_matches_numpy_goldens(name, free, c_oracle):
    """oracle/mpc_oracle.c restates the integral-form cost (fixed grid: cost x dt_ref; variable grid: dt a variable with the state-dt and
    control-dt coupling in the dt border column) against the numpy fixtures."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = R.config_unicycle_quadratic(20)
    cfg.integral_form = True
    if free:
        cfg.dt_free, cfg.dt_lb, cfg.dt_ub, cfg.xf_fixed, cfg.Qf, cfg.R = True, 0.01, 2.0, (True, True, True), None, np.array([1.0, 0.5])
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6 and np.abs(do - g["dt"]).max() < 1e-7
    assert np.abs(it - g["iters"]).max() <= 2


def test_c_oracle_footprints_and_dynamic_obstacles_match_numpy_goldens(c_oracle):
    """oracle/mpc_oracle.c restates the heading-dependent clearance rows (line / polygon / two-circle footprints) and the dt-dependent rows of
    dynamic obstacles independently of ipm_dense.py: same solutions as the numpy fixtures."""
    def check(g, cfg, obstacles, O, V, M, tol_it=2):
        xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=obstacles,
                                                  obst=c_oracle.obst_from_nlp_config(cfg, O, V, M))
        assert (st == 0).all()
        assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6 and np.abs(do - g["dt"]).max() < 1e-7
        assert np.abs(it - g["iters"]).max() <= tol_it
    # line footprint, point obstacles
    g = np.load(os.path.join(GOLD, "carlike_line_footprint_n30.npz"))
    B, O = g["pts"].shape[:2]
    cfg = R.config_carlike_min_time(30)
    cfg.footprint_kind, cfg.footprint_params = R.FOOTPRINT_LINE, tuple(g["line"])
    cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist = 0.27, 0.5, 2.5
    check(g, cfg, (np.full(B, O, np.int32), np.ones((B, O), np.int32), g["pts"].reshape(B, O, 1, 2)), O, 1, int(g["max_rows"]))
    # polygon footprint, point obstacles
    g = np.load(os.path.join(GOLD, "carlike_polygon_footprint_n30.npz"))
    B, O = g["pts"].shape[:2]
    cfg = R.config_carlike_min_time(30)
    cfg.footprint_kind, cfg.footprint_params = R.FOOTPRINT_POLYGON, tuple(g["poly"])
    cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist = 0.15, 0.5, 2.5
    check(g, cfg, (np.full(B, O, np.int32), np.ones((B, O), np.int32), g["pts"].reshape(B, O, 1, 2)), O, 1, int(g["max_rows"]))
    # two circles, polygon obstacles
    g = np.load(os.path.join(GOLD, "unicycle_two_circles_obstacles_n30.npz"))
    cfg = R.config_unicycle_quadratic(30)
    cfg.footprint_kind, cfg.footprint_params = R.FOOTPRINT_TWO_CIRCLES, tuple(g["two"])
    check(g, cfg, (g["n_obstacles"], g["n_vertices"], g["vertices"]), g["vertices"].shape[1], g["vertices"].shape[2], int(g["max_rows"]))
    # dynamic obstacles (dt free)
    g = np.load(os.path.join(GOLD, "carlike_dynamic_obstacles_n30.npz"))
    cfg = R.config_carlike_min_time(30)
    cfg.enable_dynamic_obstacles, cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist = True, 0.3, 0.5, 2.5
    check(g, cfg, (g["n_obstacles"], g["n_vertices"], g["vertices"], g["radius"], g["velocity"]), g["vertices"].shape[1], 1, int(g["max_rows"]))


def test_c_oracle_terminal_ball_matches_numpy_golden(c_oracle):
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_ball_n20.npz"))
    cfg = R.config_unicycle_quadratic(20)
    cfg.Q, cfg.R, cfg.Qf, cfg.terminal_ball_S, cfg.terminal_ball_gamma = g["Q"], g["R"], None, g["S"], float(g["gamma"])
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6
    assert np.abs(it - g["iters"]).max() <= 2


def _polish_fixture(cfg, inp, x, u, dt, **nlp_kw):
    """SLSQP on the REFERENCE-form NLP (reference row forms, teb-style distance functions, division by dt) started at a fixture's
    solution: the fixture must be feasible there and SLSQP must not find a better objective nearby."""
    nlp = R.ReferenceNlp(cfg, inp, **nlp_kw)
    z0 = nlp.pack(R.Trajectory(x, u[:-1], float(dt)))
    lb, ub = nlp.bounds()
    bnds = [(None if l < -1e29 else l, None if b > 1e29 else b) for l, b in zip(lb, ub)]
    if cfg.dt_free:
        bnds[-1] = (max(1e-3, cfg.dt_lb), cfg.dt_ub)
    cons = [{"type": "eq", "fun": nlp.equalities}, {"type": "ineq", "fun": lambda z: -nlp.inequalities(z)}]
    r = minimize(nlp.objective, z0, method="SLSQP", bounds=bnds, constraints=cons, options=dict(maxiter=40, ftol=1e-12))
    assert np.abs(nlp.equalities(z0)).max() < 1e-7 and nlp.inequalities(z0).max() < 1e-7
    f0 = nlp.objective(z0)
    ok = np.abs(nlp.equalities(r.x)).max() < 1e-7 and nlp.inequalities(r.x).max() < 1e-7
    assert (not ok) or r.fun > f0 - 1e-6 * max(1.0, abs(f0)), (r.fun, f0)
    if r.status == 0:            # SLSQP agrees that this is a solution: it must not have walked away either
        assert np.abs(r.x - z0).max() < 2e-3, np.abs(r.x - z0).max()
    _polish_fixture.last_status = r.status
    return f0


def test_late_round1_fixtures_are_kkt_points_of_the_reference_form():
    """independent pin of the fixtures behind the rows added late in round 1: via-points, terminal ball, integral form with dt free,
    line footprint, dynamic obstacles -- first instance of each, polished with SLSQP on the reference-form NLP."""
    g = np.load(os.path.join(GOLD, "carlike_via_points_n30.npz"))
    cfg = R.config_carlike_min_time(30)
    cfg.objective, cfg.vp_position_weight = R.OBJ_MIN_TIME_VIA_POINTS, float(g["wp"])
    vps = g["via"][0, :int(g["n_via"][0])]
    inp = R.CycleInputs(x0=g["x0"][0], xf=g["xf"][0], u_prev=g["u_prev"][0], dt_prev=float(g["dt_prev"][0]), via_points=vps)
    f0 = _polish_fixture(cfg, inp, g["x"][0], g["u"][0], g["dt"][0], via_idx=list(g["idx"][0, :len(vps)]))
    assert abs(f0 - float(g["objective"][0])) < 1e-9 * max(1.0, f0)

    g = np.load(os.path.join(GOLD, "unicycle_quadratic_ball_n20.npz"))
    cfg = R.config_unicycle_quadratic(20)
    cfg.Q, cfg.R, cfg.Qf, cfg.terminal_ball_S, cfg.terminal_ball_gamma = g["Q"], g["R"], None, g["S"], float(g["gamma"])
    inp = R.CycleInputs(x0=g["x0"][0], xf=g["xf"][0], u_prev=g["u_prev"][0], dt_prev=float(g["dt_prev"][0]))
    _polish_fixture(cfg, inp, g["x"][0], g["u"][0], g["dt"][0])

    g = np.load(os.path.join(GOLD, "unicycle_quadratic_integral_free_dt_n20.npz"))
    cfg = R.config_unicycle_quadratic(20)
    cfg.dt_free, cfg.dt_lb, cfg.dt_ub, cfg.xf_fixed, cfg.Qf, cfg.integral_form, cfg.R = True, 0.01, 2.0, (True, True, True), None, True, np.array([1.0, 0.5])
    inp = R.CycleInputs(x0=g["x0"][0], xf=g["xf"][0], u_prev=g["u_prev"][0], dt_prev=float(g["dt_prev"][0]))
    f0 = _polish_fixture(cfg, inp, g["x"][0], g["u"][0], g["dt"][0])
    assert abs(f0 - float(g["objective"][0])) < 1e-9 * max(1.0, f0)

    g = np.load(os.path.join(GOLD, "carlike_line_footprint_n30.npz"))
    cfg = R.config_carlike_min_time(30)
    cfg.footprint_kind, cfg.footprint_params = R.FOOTPRINT_LINE, tuple(g["line"])
    cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist = 0.27, 0.5, 2.5
    i = int(np.argmin(np.abs(g["dmin"] - 0.27)))                     # an instance with a binding row
    obs = [R.Obstacle(R.OBST_POINT, g["pts"][i, o:o + 1]) for o in range(g["pts"].shape[1])]
    inp = R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i]), obstacles=obs)
    rel, _ = R.associate_obstacles(cfg, R.cold_start(cfg, g["x0"][i], g["xf"][i]), obs, max_rows=int(g["max_rows"]))
    _polish_fixture(cfg, inp, g["x"][i], g["u"][i], g["dt"][i], relevant=rel)

    g = np.load(os.path.join(GOLD, "carlike_dynamic_obstacles_n30.npz"))
    cfg = R.config_carlike_min_time(30)
    cfg.enable_dynamic_obstacles, cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist = True, 0.3, 0.5, 2.5
    obs = [R.Obstacle(R.OBST_CIRCLE, g["vertices"][0, 0], radius=float(g["radius"][0, 0]), velocity=g["velocity"][0, 0]),
           R.Obstacle(R.OBST_POINT, g["vertices"][0, 1])]
    inp = R.CycleInputs(x0=g["x0"][0], xf=g["xf"][0], u_prev=g["u_prev"][0], dt_prev=float(g["dt_prev"][0]), obstacles=obs)
    rel, reld = R.associate_obstacles(cfg, R.cold_start(cfg, g["x0"][0], g["xf"][0]), obs, max_rows=int(g["max_rows"]) - 1)
    _polish_fixture(cfg, inp, g["x"][0], g["u"][0], g["dt"][0], relevant=rel, relevant_dyn=reld)


# ---- an INDEPENDENT solver from the reference's cold start (SURVEY 8c level 2): fixtures of tests/golden/make_cold_start_scipy.py
def _vs_independent_sqp(tag, which, solve):
    """`solve(inputs...)` -> (x, u, dt, status, iters).  Compares with the stored SLSQP results (headings modulo 2 pi: SLSQP does not wrap).
    Returns (same mask, err)."""
    import mpc_local_planner_amd.workloads as W
    from oracle import candidates as OC, kkt_check as KC
    suffix = {"3b": "_binding", "3c": "_touching"}.get(which, "")       # config 3 on placements where the clearance rows bind (make_cold_start_scipy.py 3b / 3c)
    g = np.load(os.path.join(GOLD, f"cold_start_scipy_config{str(which)[0]}{suffix}.npz"))
    K = int(g["count"])
    if which == 2:
        inputs = W.carlike_min_time_inputs(K); obst = None
        ocfg = R.config_carlike_min_time(50)
    else:
        lat = tuple(float(v) for v in g["lateral"]) if "lateral" in g.files else (0.3, 1.5)
        x0, xf, up, dtp, obst = W.unicycle_obstacle_inputs(K, n_obst=16, max_vertices=6, lateral=lat)
        inputs = (x0, xf, up, dtp)
        ocfg = R.config_unicycle_quadratic(80)
    x, u, dt, st, it = solve(ocfg, inputs, obst)
    d = x - g["x"]; d[..., 2] = OC.wrap(d[..., 2])
    err = np.abs(d).reshape(K, -1).max(1)
    valid = g["violation"] < 1e-6         # (on the hardest placement SLSQP itself ends infeasible in some instances: those have no reference point)
    conv = (st == 0) & valid
    same = conv & (err < 2e-4)            # SLSQP works with finite-difference gradients: its own accuracy is ~1e-5 .. 1e-4
    other = np.nonzero(conv & ~same)[0]
    res = KC.kkt_many(ocfg, inputs[0], inputs[1], inputs[2], inputs[3], x, u, dt, other, obstacles=obst, max_rows=4 if obst is not None else None)
    bad = [i for i in other if not KC.is_kkt_point(res[i])]
    dobj = [res[i]["objective"] - float(g["objective"][i]) for i in other]
    print(f"[{tag}] {K} instances from the reference cold start ({int(valid.sum())} with a feasible SLSQP result): solver converged {int(conv.sum())} of those, SLSQP violation <= {g['violation'][valid].max():.1e}; "
          f"same KKT point (<2e-4) {int(same.sum())} (median {np.median(err[same]):.1e}); different local optimum {len(other)} "
          f"(all KKT points of the reference-form NLP: {not bad}; objective solver - SLSQP: "
          + (f"min {min(dobj):+.3f} median {np.median(dobj):+.3f} max {max(dobj):+.3f}, solver better in {sum(d < 0 for d in dobj)}" if dobj else "-") + ")")
    assert valid.sum() >= 0.5 * K and not bad
    return same, err, conv


def _c_solve(c_oracle):
    def solve(ocfg, inputs, obst):
        if obst is None:
            return c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *inputs)
        return c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *inputs, obstacles=obst, obst=c_oracle.obst_from_nlp_config(ocfg, 16, 6, 4))
    return solve


def test_independent_sqp_from_the_cold_start_config3(c_oracle):
    """config 3 (quadratic form, 16 polygon obstacles, n = 80): a near-convex problem -- an active-set SQP started at the reference's cold
    start and the interior-point oracle land on the SAME point in every one of the 32 instances."""
    same, err, conv = _vs_independent_sqp("C oracle vs SLSQP, config 3", 3, _c_solve(c_oracle))
    assert conv.all() and same.all()


@pytest.mark.parametrize("which", ["3b", "3c"])
def test_independent_sqp_from_the_cold_start_config3_with_binding_rows(c_oracle, which):
    """config 3 where the clearance rows BIND.  3b = the bench leg's placement (polygons 0.15 .. 0.8 m beside the start-goal line, d_min 0.2: rows start
    violated by up to 5 cm, about half of the solutions end with an active row); 3c = polygons reaching to within 2 cm of the line (rows start violated by up to 18 cm: detours).
    SLSQP starts at the reference's cold start with the rows associated on it, like the interior-point solve.  Every converged interior-point result is at SLSQP's point or is
    a KKT point of the reference-form NLP in its own right (checked inside the helper); the shares are asserted as measured."""
    same, err, conv = _vs_independent_sqp(f"C oracle vs SLSQP, config {which}", which, _c_solve(c_oracle))
    K = len(conv)
    if which == "3b":
        assert conv.sum() >= K - 2 and same.sum() >= 0.8 * conv.sum()
    else:
        assert conv.sum() >= 0.3 * K and same.sum() >= 0.5 * conv.sum()


def test_independent_sqp_from_the_cold_start_config2(c_oracle):
    """config 2 (car-like minimum time, n = 50): the NLP has many local optima (driving-direction reversals), so two different solvers that
    start at the same cold start agree only where their iterates stay in the same basin -- 16 of 32 here; in the other instances BOTH end at
    KKT points of the reference-form NLP (the interior-point result is checked with oracle/kkt_check.py, SLSQP reports success), with the
    better objective on either side.  This is what "parity with the reference's Ipopt" can mean for this workload: the same NLP, KKT points
    of it, and identical results wherever the iterate paths coincide -- not a solver-independent answer."""
    same, err, conv = _vs_independent_sqp("C oracle vs SLSQP, config 2", 2, _c_solve(c_oracle))
    assert conv.sum() >= 28 and same.sum() >= 14          # r04 (adaptive barrier rule, rate-feasible control seed): 29 converged, 16 at SLSQP's point (r03: 30 / 14)


def test_numpy_and_c_oracle_agree_on_unfiltered_config2_instances(c_oracle):
    """The golden fixtures keep well-behaved instances only (make_golden.py drops what needs > 45 iterations).  Here the first 16 instances of
    the config-2 distribution AND the first 8 slow ones (> 45 iterations) among the first 256 go through both oracles as they come:
    dense numpy linear algebra and the banded-LU C restatement must produce the same status and, where converged, the same trajectory -- or, on a
    slow instance where a line-search tie flips between the two (one of 24 here, 93 iterations), two trajectories that are BOTH KKT points of the
    reference-form NLP."""
    from oracle import kkt_check as KC
    import mpc_local_planner_amd.workloads as W
    n, K = 50, 256
    x0, xf, up, dtp = W.carlike_min_time_inputs(K)
    ocfg = R.config_carlike_min_time(n)
    o = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    slow = [i for i in range(K) if o[4][i] > 45][:8]           # the first slow ones (one of them hits the iteration cap) ...
    pick = sorted(set(list(range(16)) + slow))                # ... next to the first 16 as they come
    hard = parted = 0
    for i in pick:
        inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
        r = I.solve(ocfg, inp, R.cold_start(ocfg, x0[i], xf[i]), opt=I.IpmOptions(max_iter=100))
        if r.status != o[3][i]:
            assert o[4][i] > 45 or r.iters > 45, (i, r.status, o[3][i])      # a slow run may end on either side of the cap
            parted += 1
        elif r.status == 0:
            err = max(np.abs(r.traj.x - o[0][i]).max(), np.abs(r.traj.u - o[1][i][:-1]).max(), abs(r.traj.dt - o[2][i]))
            if err >= 1e-5:                                     # (flat directions: two round-off paths to the same minimum-time solution differ by ~1e-6 in the states)
                parted += 1
                assert o[4][i] > 45, i                              # only a slow instance may part ways
                assert KC.is_kkt_point(KC.kkt_residuals(ocfg, x0[i], xf[i], up[i], dtp[i], r.traj.x, r.traj.u, r.traj.dt)), i
                assert KC.is_kkt_point(KC.kkt_residuals(ocfg, x0[i], xf[i], up[i], dtp[i], o[0][i], o[1][i], o[2][i])), i
        hard += int(o[4][i] > 45)
    assert hard >= 6 and parted <= 3, (hard, parted)


def test_terminal_cost_applies_to_the_minimum_time_objective_too(c_oracle):
    """planning/terminal_cost is configured independently of planning/objective (src/controller.cpp:641-672) and its edge exists whenever the final
    state is not completely fixed (finite_differences_grid_se2.cpp:128-133): minimum time + quadratic terminal cost + free final state.
    numpy dense and C oracle agree, and the results are KKT points of the reference-form NLP."""
    import dataclasses
    import mpc_local_planner_amd.workloads as W
    from oracle import kkt_check as KC
    ocfg = dataclasses.replace(R.config_carlike_min_time(20), Qf=np.array([2.0, 2.0, 2.0]), xf_fixed=(False, False, False), dt_lb=0.05)
    x0, xf, up, dtp = W.carlike_min_time_inputs(8, seed=5, goal_range=(1.0, 2.5))
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    assert (st == 0).all()
    away = np.abs(xo[:, -1] - xf).max(1)
    assert away.max() > 0.05          # the final state is free: the terminal cost trades distance to the goal against time
    for i in range(4):
        inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
        d = I.solve(ocfg, inp, R.cold_start(ocfg, x0[i], xf[i]))
        assert d.status == 0 and np.abs(d.traj.x - xo[i]).max() < 1e-6
        k = KC.kkt_residuals(ocfg, x0[i], xf[i], up[i], dtp[i], xo[i], uo[i], do[i])
        assert KC.is_kkt_point(k), k
        # and it is NOT a KKT point of the problem without the terminal cost
        k0 = KC.kkt_residuals(dataclasses.replace(ocfg, Qf=None), x0[i], xf[i], up[i], dtp[i], xo[i], uo[i], do[i])
        assert k0["stat"] > 1e-3


COST_VARIANTS = ["full_weights", "trapezoid_fixed_dt", "trapezoid_free_dt", "trapezoid_xf_fixed_free_dt", "hybrid", "hybrid_integral", "all"]


def cost_variant(name, n=16):
    """full Q / R / Qf / S (src/controller.cpp:561-592,652-668,686-702), trapezoidal rule for integral-form costs (finite_differences_grid_se2.cpp:63-68),
    hybrid minimum time + control cost (src/controller.cpp:616-618) on the unicycle quadratic-form example"""
    import dataclasses
    FQ = np.array([[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.4]]); FR = np.array([[0.1, 0.02], [0.02, 0.05]])
    FQF = np.array([[8.0, 1.0, 0.0], [1.0, 9.0, 0.5], [0.0, 0.5, 0.6]]); FS = np.array([[1.0, 0.2, 0.0], [0.2, 1.0, 0.1], [0.0, 0.1, 0.5]])
    base = R.config_unicycle_quadratic(n)
    free = dict(dt_free=True, dt_lb=0.05, dt_ub=1.0)
    hyb = dict(Q=np.zeros(3), Qf=None, hybrid_min_time=True, dt_free=True, xf_fixed=(True, True, True), R=np.array([1.0, 0.5]))
    return {
        "full_weights": dataclasses.replace(base, Q=FQ, R=FR, Qf=FQF),
        "trapezoid_fixed_dt": dataclasses.replace(base, integral_form=True, cost_integration="trapezoidal_rule"),
        "trapezoid_free_dt": dataclasses.replace(base, integral_form=True, cost_integration="trapezoidal_rule", **free),
        "trapezoid_xf_fixed_free_dt": dataclasses.replace(base, integral_form=True, cost_integration="trapezoidal_rule", xf_fixed=(True, True, True), **free),
        "hybrid": dataclasses.replace(base, **hyb),
        "hybrid_integral": dataclasses.replace(base, integral_form=True, **hyb),
        "all": dataclasses.replace(base, Q=FQ, R=FR, Qf=FQF, integral_form=True, cost_integration="trapezoidal_rule", terminal_ball_S=FS, terminal_ball_gamma=0.3, **free),
    }[name]


@pytest.mark.parametrize("name", COST_VARIANTS)
def test_c_oracle_cost_variants_match_numpy_and_are_kkt_points_of_the_reference_form(name, c_oracle):
    import mpc_local_planner_amd.workloads as W
    from oracle import kkt_check as KC
    ocfg = cost_variant(name)
    x0, xf, up, dtp = W.unicycle_quadratic_inputs(6, seed=11)
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    assert (st == 0).all()
    for i in range(2):
        inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
        d = I.solve(ocfg, inp, R.cold_start(ocfg, x0[i], xf[i]))
        assert d.status == 0 and np.abs(d.traj.x - xo[i]).max() < 1e-6 and abs(d.traj.dt - do[i]) < 1e-6
        k = KC.kkt_residuals(ocfg, x0[i], xf[i], up[i], dtp[i], xo[i], uo[i], do[i])
        assert KC.is_kkt_point(k), (name, k)
    if name == "trapezoid_free_dt":       # the rule matters: the left-sum solution differs
        import dataclasses
        xl = c_oracle.solve_batch(c_oracle.from_nlp_config(dataclasses.replace(ocfg, cost_integration="left_sum")), x0, xf, up, dtp)[0]
        assert np.abs(xl - xo).max() > 1e-3


@pytest.mark.parametrize("name", [v for v in COST_VARIANTS if v != "trapezoid_xf_fixed_free_dt"])
def test_cost_variants_independent_sqp_from_the_cold_start(name, c_oracle):
    """SURVEY 8c level 2 for the cost variants: scipy's SLSQP (an active-set SQP with no code in common with the interior-point implementations), started
    at the reference's cold start on the reference-form NLP, ends at the interior-point oracle's point (these quadratic-form problems are near-convex).
    Not for the variant with a FIXED goal and a free dt: like the minimum-time problem it has several local optima and the two solvers settle in different
    ones (both KKT points: test_c_oracle_cost_variants_match_numpy_and_are_kkt_points_of_the_reference_form)."""
    from scipy.optimize import minimize
    import mpc_local_planner_amd.workloads as W
    ocfg = cost_variant(name, 12)
    x0, xf, up, dtp = W.unicycle_quadratic_inputs(3, seed=11)
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    assert (st == 0).all()
    for i in range(3):
        inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
        nlp = R.ReferenceNlp(ocfg, inp)
        z0 = nlp.pack(I.controls_from_states(ocfg, R.cold_start(ocfg, x0[i], xf[i])))
        lb, ub = nlp.bounds()
        z0 = np.minimum(np.maximum(z0, lb), ub)
        bnds = [(None if l < -1e29 else l, None if u > 1e29 else u) for l, u in zip(lb, ub)]
        cons = [{"type": "eq", "fun": nlp.equalities}, {"type": "ineq", "fun": lambda z: -nlp.inequalities(z)}]
        r = minimize(nlp.objective, z0, method="SLSQP", bounds=bnds, constraints=cons, options=dict(maxiter=600, ftol=1e-13))
        t = nlp.unpack(r.x)
        assert np.abs(t.x - xo[i]).max() < 1e-4 and abs(t.dt - do[i]) < 1e-4, (name, i, np.abs(t.x - xo[i]).max())
        assert abs(r.fun - nlp.objective(nlp.pack(R.Trajectory(xo[i], uo[i][:ocfg.n - 1], float(do[i]))))) < 1e-6 * max(1.0, abs(r.fun))


def test_near_goal_stall_and_the_acceptable_level_stop():
    """A 4-point grid 0.27 m in front of the goal (a cycle of the `carlike_to_the_goal` closed loop of the reference's plugin).  With tol 1e-8 and
    WITHOUT Ipopt's acceptable-level stop the solve stands at 6e-8 after 13 iterations (1.2e-8 after 15 with the monotone barrier rule): the terminal rows'
    dual regularisation leaves a residual the Newton step cannot remove, the merit function's predicted decrease counts on removing it, and the line search
    refuses what follows -- the solve ends in a line-search failure (monotone rule: at max_iter).  At tol 1e-5 it converges; with the stop (the default of all
    three solvers: IpmOptions.acceptable_tol / oracle_config / mpc_config) it ends at that 13th iterate with status 0, and the answers agree to 5e-6.
    This is the l1-MERIT line search (mpc_config.line_search = MPC_LS_MERIT, the default of rounds 1-5); the filter line search (the default since r06) does not
    stall here: it reaches tol 1e-8 with the stop switched off (last lines)."""
    cfg = R.config_carlike_min_time(4)
    inp = R.CycleInputs(x0=np.array([1.836, 0.676, 0.366]), xf=np.array([2.087, 0.769, 0.2927]), u_prev=np.array([0.4, 0.0]), dt_prev=0.1)
    stalled = _ipm(cfg, inp, acceptable_tol=0.0, globalization="merit")
    loose = _ipm(cfg, inp, tol=1e-5, globalization="merit")
    stopped = _ipm(cfg, inp, globalization="merit")
    mono = _ipm(cfg, inp, mu_strategy="monotone", globalization="merit")
    mono_stalled = _ipm(cfg, inp, mu_strategy="monotone", acceptable_tol=0.0, globalization="merit")
    # the stall sits at the rounding level of the merit function: on this container's BLAS it is there; the C solver's test below, whose arithmetic does
    # not depend on the machine, is the one that insists on it
    assert stalled.status in (0, 1, 2) and stalled.kkt_error < 1e-4
    if stalled.status != 0:
        assert min([stalled.kkt_error] + [h["e0"] for h in stalled.history]) < 1e-7          # it had been there
    assert mono_stalled.status in (0, 1) and (mono_stalled.status == 0 or (mono_stalled.iters == 100 and min(h["e0"] for h in mono_stalled.history) < 2e-8))
    assert loose.status == 0 and loose.iters <= 16
    assert stopped.status == 0 and stopped.iters <= 16 and stopped.kkt_error < 1e-7
    assert mono.status == 0 and mono.iters <= 16 and mono.kkt_error < 2e-8
    for other in (stalled, loose, mono):
        assert np.abs(stopped.traj.x - other.traj.x).max() < 5e-6
        assert np.abs(stopped.traj.u - other.traj.u).max() < 5e-6
        assert abs(stopped.traj.dt - other.traj.dt) < 5e-6
    # Ipopt's defaults
    assert I.IpmOptions().acceptable_tol == 1e-6 and I.IpmOptions().acceptable_iter == 15 and I.IpmOptions().globalization == "filter"
    filt = _ipm(cfg, inp, acceptable_tol=0.0)
    assert filt.status == 0 and filt.iters <= 20 and filt.kkt_error <= 1e-8 and np.abs(filt.traj.x - stopped.traj.x).max() < 5e-6


def test_near_goal_stall_in_the_c_solver_and_its_acceptable_level_stop(c_oracle):
    """The same instance in the C solver under the l1 merit (oracle_config.line_search = 0): no success with the rule switched off (acceptable_tol < 0), 13 iterations with
    it (the default), same point as numpy; under the filter (the default line search) the instance converges to tol without the rule; and the headline workload is untouched by the rule: bit-identical trajectories, statuses and iteration counts on 256 cold starts."""
    cfg = R.config_carlike_min_time(4)
    x0, xf = np.array([[1.836, 0.676, 0.366]]), np.array([[2.087, 0.769, 0.2927]])
    up, dtp = np.array([[0.4, 0.0]]), np.array([0.1])
    off = c_oracle.from_nlp_config(cfg, max_iter=100, tol=1e-8, acceptable_tol=-1.0, line_search=0)
    on = c_oracle.from_nlp_config(cfg, max_iter=100, tol=1e-8, line_search=0)
    ref = _ipm(cfg, R.CycleInputs(x0=x0[0], xf=xf[0], u_prev=up[0], dt_prev=0.1), globalization="merit")
    r = c_oracle.solve_batch(off, x0, xf, up, dtp, nthreads=1)
    assert r[3][0] == 2 and r[4][0] <= 20           # line search failure at the stall (adaptive barrier rule, the default)
    r = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, max_iter=100, tol=1e-8, acceptable_tol=-1.0, mu_strategy=1, line_search=0), x0, xf, up, dtp, nthreads=1)
    assert r[3][0] == 1 and r[4][0] == 100          # the monotone rule creeps on to max_iter
    r = c_oracle.solve_batch(on, x0, xf, up, dtp, nthreads=1)
    assert r[3][0] == 0 and r[4][0] == ref.iters
    assert np.abs(r[0][0] - ref.traj.x).max() < 1e-7 and np.abs(r[1][0][:3] - ref.traj.u).max() < 1e-7 and abs(r[2][0] - ref.traj.dt) < 1e-7
    # a looser level with the counting half: the run ends early, by whichever rule fires first
    r = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, max_iter=100, tol=1e-8, acceptable_tol=1e-5, acceptable_iter=15, line_search=0), x0, xf, up, dtp, nthreads=1)
    assert r[3][0] == 0 and r[4][0] <= 30
    r = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, max_iter=100, tol=1e-8, acceptable_tol=-1.0), x0, xf, up, dtp, nthreads=1)
    assert r[3][0] == 0 and r[4][0] <= 20           # the filter line search: no stall
    # headline workload (config 2): the rule changes nothing
    from mpc_local_planner_amd import workloads as W
    cfg2 = R.config_carlike_min_time(50)
    w = W.carlike_min_time_inputs(256)
    a = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg2, max_iter=100, tol=1e-8, acceptable_tol=-1.0), *w)
    b = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg2, max_iter=100, tol=1e-8), *w)
    for i in range(5):
        assert np.array_equal(a[i], b[i])


def test_kkt_checker_takes_the_clearance_rows_of_the_trajectory_a_solve_started_from(c_oracle):
    """The clearance rows of a solve are the ones associated on the trajectory it STARTS from (StageInequalitySE2::update runs in the grid update, before the solve; the
    product and the C oracle do the same).  A solve that starts from a candidate initial trajectory therefore carries that trajectory's rows: its result is a KKT point of the
    NLP with THOSE rows (oracle/kkt_check.py, start_x), and need not be one of the NLP with the rows of the reference's cold start.  Workload: the dynamic-obstacle + line-footprint
    batch of tests/test_gpu_ext_rows.py, solved by the C oracle from a Hermite seed (what a hedge of the product does); found on the MI355X in round 3, where three hedge answers
    failed the check against the cold start's rows (stationarity 1e-2) and pass with their own (1e-8)."""
    import mpc_local_planner_amd.workloads as W
    from oracle import candidates as OC, kkt_check as KC
    B, n = 24, 50
    x0, xf, up, dtp = W.carlike_min_time_inputs(128, seed=931, goal_range=(2.0, 5.0))
    rng = np.random.default_rng(932)          # point_obstacles() of tests/test_gpu_ext_rows.py
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (128, 3, 1)) * d + rng.uniform(0.6, 1.1, (128, 3, 1)) * rng.choice([-1.0, 1.0], (128, 3, 1)) * nrm
    no, nv, vt = np.full(128, 3, np.int32), np.ones((128, 3), np.int32), pts.reshape(128, 3, 1, 2)
    rad = np.zeros((128, 3)); vel = np.zeros((128, 3, 2))
    d2 = xf[:, :2] - x0[:, :2]
    nr2 = np.stack([-d2[:, 1], d2[:, 0]], -1) / np.linalg.norm(d2, axis=-1, keepdims=True)
    vt[:, 0, 0] = x0[:, :2] + 0.5 * d2 + 1.2 * nr2; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nr2
    pick = np.array([66, 86, 118, 7] + list(range(20)))[:B]           # the three instances of the GPU finding first
    x0, xf, up, dtp = x0[pick], xf[pick], up[pick], dtp[pick]
    obs = tuple(a[pick] for a in (no, nv, vt, rad, vel))
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params = 2, (0.0, 0.0, 0.4, 0.0)
    ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, 0.27, 0.5, 2.5
    seed = OC.guess(5, x0, xf, n, ocfg.dt_ref, param=2.0)
    x, u, dt, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, init=seed, obstacles=obs, obst=c_oracle.obst_from_nlp_config(ocfg, 3, 1, 4))[:5]
    conv = [int(i) for i in np.nonzero(st == 0)[0]]
    assert len(conv) >= B // 2
    own = KC.kkt_many(ocfg, x0, xf, up, dtp, x, u, dt, conv, obstacles=obs, max_rows=4, start_x=seed[0])
    cold = KC.kkt_many(ocfg, x0, xf, up, dtp, x, u, dt, conv, obstacles=obs, max_rows=4)
    ok_own = [i for i in conv if KC.is_kkt_point(own[i], 1e-6, 1e-6, 1e-6)]
    not_cold = [i for i in conv if not KC.is_kkt_point(cold[i], 1e-6, 1e-6, 1e-6)]
    print(f"[rows of the start trajectory] {len(conv)} of {B} converge from the Hermite seed; KKT points of the NLP with the seed's rows: {len(ok_own)}; of those NOT KKT points with the "
          f"cold start's rows: {len(not_cold)} (worst stationarity there {max([cold[i]['stat'] for i in not_cold], default=0):.1e})")
    assert len(ok_own) == len(conv)
    assert len(not_cold) >= 1          # the distinction is real on this workload


@pytest.mark.parametrize("ls", ["filter", "merit"])
def test_restoration_for_jammed_clearance_rows_in_both_cpu_solvers(c_oracle, ls):
    """r05: clearance rows that jam -- five iterations in a row whose fraction-to-boundary limit on the primal step is below 0.05 with the infeasibility still at
    80 % -- turn elastic (g + s - e = 0, e >= 0, + 1000 e).  Car-like minimum time, n = 30, three point obstacles 0.05 .. 0.5 m beside the path (d_min 0.3: most rows start violated),
    64 instances, reference path alone.  (i) The mode changes the outcome of some instances and ONLY helps the converged count: instances that ran into the iteration limit converge.
    (ii) Along the restoration path the dense numpy solver and the banded-LU C solver -- two implementations of the rule, two linear algebras -- produce the same iterate sequence:
    same iteration counts, trajectories equal to 1e-6 (they are equal to 1e-13 on most).  (iii) What is returned is a KKT point of the reference-form NLP with the frozen rows
    (oracle/kkt_check.py), i.e. the elastic variables ended at zero.
    (i) is the statement for the l1-merit line search the mode was built under (r05).  Under the filter line search (the default since r06) this small workload converges 63 of 64
    either way (one instance each way ends at the iteration limit); what the mode is worth there shows on the harder workloads (moving obstacle + polygon footprint: 58 % without,
    93 % with) -- here the filter half checks (ii) and (iii) and that the converged count does not fall."""
    import ctypes as C
    from oracle import kkt_check as KC
    import mpc_local_planner_amd.workloads as W
    B, n, O = 64, 30, 3
    x0, xf, up, dtp = W.carlike_min_time_inputs(B, seed=77, goal_range=(2.0, 4.0))
    rng = np.random.default_rng(78)
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, O, 1)) * d + rng.uniform(0.05, 0.5, (B, O, 1)) * rng.choice([-1.0, 1.0], (B, O, 1)) * nrm
    obstacles = (np.full(B, O, np.int32), np.ones((B, O), np.int32), pts.reshape(B, O, 1, 2))
    ocfg = R.config_carlike_min_time(n)
    ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = 0.3, 0.5, 2.5
    ob = c_oracle.obst_from_nlp_config(ocfg, O, 1, 4)
    lib = c_oracle._load()
    on = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, line_search=0 if ls == "merit" else 1), x0, xf, up, dtp, obstacles=obstacles, obst=ob)
    try:
        lib.oracle_set_algo(C.c_int(10), C.c_double(0.0))                  # experiment switch: restoration off
        off = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, line_search=0 if ls == "merit" else 1), x0, xf, up, dtp, obstacles=obstacles, obst=ob)
    finally:
        lib.oracle_set_algo(C.c_int(10), C.c_double(1000.0))
    changed = np.flatnonzero((on[3] != off[3]) | (on[4] != off[4]))
    print(f"[restoration, CPU, {ls}] converged without / with: {int((off[3] == 0).sum())} / {int((on[3] == 0).sum())} of {B}; instances whose iterate path it changes: "
          f"{[(int(i), int(off[3][i]), int(off[4][i]), int(on[3][i]), int(on[4][i])) for i in changed]}")
    assert len(changed) >= 4
    if ls == "merit":
        assert (on[3] == 0).sum() >= (off[3] == 0).sum() + 2
        assert not ((off[3] == 0) & (on[3] != 0)).any()                    # nothing that converged without the mode is lost with it
    else:
        assert (on[3] == 0).sum() >= (off[3] == 0).sum() and (on[3] == 0).sum() >= B - 1
    same_it, checked = 0, 0
    for i in [int(i) for i in changed if on[3][i] == 0][:5]:
        obs = [R.Obstacle(R.OBST_POINT, pts[i, o:o + 1]) for o in range(O)]
        inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]), obstacles=obs)
        init = R.cold_start(ocfg, x0[i], xf[i])
        rel, _ = R.associate_obstacles(ocfg, init, obs, max_rows=4)
        ref = I.solve(ocfg, inp, init, relevant=rel, opt=I.IpmOptions(max_iter=100, globalization=ls))
        assert ref.status == 0
        assert np.abs(ref.traj.x - on[0][i]).max() < 1e-4 and abs(ref.traj.dt - on[2][i]) < 1e-8
        same_it += int(ref.iters == on[4][i]); checked += 1
    assert checked >= 3 and same_it >= checked - 1
    res = KC.kkt_many(ocfg, x0, xf, up, dtp, on[0], on[1], on[2], [int(i) for i in changed if on[3][i] == 0], obstacles=obstacles, max_rows=4)
    assert all(KC.is_kkt_point(r, 1e-6, 1e-6, 1e-6) for r in res.values()), res


def test_clearance_to_every_obstacle_needs_a_renewed_association(c_oracle):
    """The clearance rows of ONE solve are those associated on the trajectory it starts from (StageInequalitySE2::update runs in the grid update, before the solve:
    stage_inequality_se2.cpp:50-162) -- in the reference as here.  Measured with plain geometry (no solver quantity) on config 3 with polygons 0.15 .. 0.8 m beside the
    path: after one solve from the cold start about one converged trajectory in five is closer than min_obstacle_dist to a polygon it carried no row for; a second solve
    that re-associates on that solution (the next outer OCP iteration / control cycle) leaves at most a stray one, a third none.  tests/test_gpu_clearance.py asserts
    the same through mpc_step_batch on the device."""
    import mpc_local_planner_amd.workloads as W
    n, B, O, V, M, dmin = 80, 96, 16, 6, 4, 0.2
    x0, xf, up, dtp, (no, nv, verts) = W.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=(0.15, 0.8))
    ocfg = R.config_unicycle_quadratic(n)
    oc, ob = c_oracle.from_nlp_config(ocfg), c_oracle.obst_from_nlp_config(ocfg, O, V, M)

    def clearance(x):
        out = np.full(B, np.inf)
        for b in range(B):
            p = x[b, 1:-1, :2]
            for o in range(int(no[b])):
                k = int(nv[b, o]); a = verts[b, o, :k]; c = np.roll(a, -1, axis=0); ab = c - a
                t = np.clip(((p[:, None, :] - a[None]) * ab[None]).sum(-1) / (ab * ab).sum(-1)[None], 0.0, 1.0)
                out[b] = min(out[b], float(np.sqrt(((p[:, None, :] - (a[None] + t[..., None] * ab[None])) ** 2).sum(-1)).min()))
        return out
    o = c_oracle.solve_batch(oc, x0, xf, up, dtp, obstacles=(no, nv, verts), obst=ob)
    share = []
    for rep in range(3):
        ok = o[3] == 0
        c = clearance(o[0])
        share.append(float(np.mean(c[ok] >= dmin - 1e-6)))
        assert ok.mean() >= 0.9
        o = c_oracle.solve_batch(oc, x0, xf, up, dtp, init=(o[0], o[1], o[2]), obstacles=(no, nv, verts), obst=ob)
    assert 0.6 < share[0] < 0.98 and share[1] >= 0.97 and share[2] == 1.0, share


def test_stage_structured_quasi_newton_hessian_experiment(c_oracle):
    """`hessian_approximation: limited-memory` of the shipped car-like file (cfg/carlike/mpc_local_planner_params.yaml:91-95) is mapped to MPC_HESSIAN_CONVEXIFIED.  The C oracle
    also carries a TRUE quasi-Newton mode that keeps the stage structure (oracle_config.hessian_mode = 2: one secant-updated 4 x 4 block per collocation increment, Griewank-Toint
    partitioned updating) so that the mapping is a measured choice: at the file's tol 1e-4 the convexified exact Hessian converges for more instances in fewer iterations than
    symmetric rank-one blocks, and damped BFGS blocks -- positive semidefinite, while the element Hessian [H_qq H_qd; H_qd' 0] is indefinite -- fail for most."""
    import ctypes as C
    import mpc_local_planner_amd.workloads as W
    B, n = 256, 50
    ocfg = R.config_carlike_min_time(n)
    inputs = W.carlike_min_time_inputs(B)
    lib = c_oracle._load()
    res = {}
    for name, mode, sr1 in (("convexified", 1, 1), ("sr1", 2, 1), ("bfgs", 2, 0)):
        lib.oracle_set_algo(C.c_int(7), C.c_double(sr1))
        try:
            o = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, tol=1e-4, hessian_mode=mode), *inputs)
        finally:
            lib.oracle_set_algo(C.c_int(7), C.c_double(1))
        res[name] = (float((o[3] == 0).mean()), float(o[4].mean()), o)
    print({k: v[:2] for k, v in res.items()})
    # r04 with the inertia test (Ipopt's): 96 % / 49 % / 15 % -- symmetric rank-one blocks are indefinite, their factorisations fail the inertia test and drown in delta_w (they
    # converged for 91 % under the inertia-free curvature test of r01-r03); the positive semidefinite BFGS blocks cannot approach the indefinite element Hessian, as before
    assert res["convexified"][0] >= 0.95 and res["convexified"][0] > max(res["sr1"][0], res["bfgs"][0]) + 0.2
    assert res["convexified"][1] < min(res["sr1"][1], res["bfgs"][1])
    # where both converge to the same basin the quasi-Newton answer is the exact-Hessian one (same NLP, same KKT points)
    a, b = res["convexified"][2], res["sr1"][2]
    both = (a[3] == 0) & (b[3] == 0)
    same = np.abs(a[0] - b[0]).reshape(B, -1).max(1)[both] < 1e-2
    assert same.mean() > 0.5


def test_converged_answers_are_local_minima_and_the_curvature_test_s_were_not(c_oracle):
    """r04: what `converged` means.  oracle/kkt_check.py::second_order measures, with differences of the reference-form NLP's functions only, the smallest eigenvalue of the
    Lagrangian's Hessian on the tangent space of the active rows of a KKT point.  With the inertia-free curvature test of r01-r03 (oracle_set_algo(9, 0)) the Newton iteration
    converges to SADDLE points -- a feasible direction of negative curvature, verified by moving along it in DESIGN section 3.2 -- for about one converged config-2 answer in
    five; with Ipopt's inertia test (the algorithm since r04: delta_w is raised until the KKT matrix has n positive and m negative eigenvalues) every converged answer of the
    sample is a minimum, in fewer iterations."""
    import ctypes as C
    import mpc_local_planner_amd.workloads as W
    B, n = 10, 50
    ocfg = R.config_carlike_min_time(n)
    pick = [0, 4, 11, 15, 24, 30, 1, 2, 3, 5]          # of the 96-instance draw: six of the 17 saddle points the curvature test returns there, four others
    x0, xf, up, dtp = (a[pick] for a in W.carlike_min_time_inputs(96))
    lib = c_oracle._load()
    out = {}
    for mode in (0, 1):
        lib.oracle_set_algo(C.c_int(9), C.c_double(mode))
        try:
            x, u, dt, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
        finally:
            lib.oracle_set_algo(C.c_int(9), C.c_double(1))
        so = K.second_order_many(ocfg, x0, xf, up, dtp, x, u, dt, np.nonzero(st == 0)[0])
        eig = np.array([r["min_eig_s"] for r in so.values()])
        out[mode] = (int((st == 0).sum()), float(it.mean()), eig, [r["n_weak"] for r in so.values()])
        print(f"inertia test {mode}: converged {out[mode][0]} / {B}, mean iterations {out[mode][1]:.1f}, smallest reduced-Hessian eigenvalues {np.sort(eig)[:5].round(4).tolist()}, "
              f"saddle points {(eig < -1e-6).sum()}")
    assert (out[0][2] < -1e-3).sum() >= 2                     # the curvature test's answers: saddle points among them
    assert (out[1][2] > -1e-6).all() and max(out[1][3]) == 0   # the inertia test's: minima (no weakly active row blurs the statement)
    assert out[1][0] >= out[0][0] and out[1][1] < out[0][1]


def _same_point_mod_2pi(xa, xb, tol):
    d = xa - xb
    d[..., 2] = (d[..., 2] + np.pi) % (2 * np.pi) - np.pi          # SLSQP does not wrap headings: compare them modulo 2 pi
    return np.abs(d).reshape(d.shape[0], -1).max(1) < tol


def test_config5_shape_vs_slsqp(c_oracle):
    """ The config-5 SHAPE (kinematic bicycle, n = 120, goals 5 .. 40 m) against an independent solver from the same cold start:
    tests/golden/cold_start_scipy_config5.npz (scipy SLSQP on the reference-form NLP, 8 instances, ~1-10 min each).  SLSQP succeeds on 6 of 8; the C oracle's reference path
    converges on all 8; on 4 of SLSQP's 6 the two end at the SAME point (headings modulo 2 pi, 1e-5; 5 of 6 under the l1-merit line search, checked too), the others are other
    local minima of this multi-modal NLP (travel times 12 ms and 14.1 s above SLSQP's); where SLSQP gives up, the interior-point answer is feasible with a lower travel time than
    SLSQP's last iterate."""
    from mpc_local_planner_amd import workloads as W
    g = np.load(os.path.join(GOLD, "cold_start_scipy_config5.npz"))
    K = int(g["count"])
    x0, xf, up, dtp = W.bicycle_min_time_inputs(K)
    cfg = R.config_bicycle_min_time(120)
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, max_iter=100), x0, xf, up, dtp)
    ok = g["success"].astype(bool)
    assert ok.sum() >= 6 and (st == 0).sum() >= 7
    same = _same_point_mod_2pi(xo, g["x"], 1e-5) & (st == 0) & ok
    obj = (cfg.n - 1) * do
    print(f"[config-5 shape vs SLSQP] SLSQP succeeded on {int(ok.sum())} of {K}; the C oracle converged on {int((st == 0).sum())}; same point (headings mod 2 pi) on {int(same.sum())} of SLSQP's; "
          f"travel time oracle - SLSQP where both have an answer: {np.round((obj - g['objective'])[ok & (st == 0)], 4).tolist()}")
    assert same.sum() >= 4
    assert np.abs(obj[same] - g["objective"][same]).max() < 1e-5
    xm, _, dm, stm, _ = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, max_iter=100, line_search=0), x0, xf, up, dtp)
    assert (_same_point_mod_2pi(xm, g["x"], 1e-5) & (stm == 0) & ok).sum() >= 5
    both = ok & (st == 0) & ~same
    assert (g["violation"][ok] < 1e-8).all()
    for i in np.where(~ok & (st == 0))[0]:          # SLSQP gave up: our answer must at least be feasible and not worse than where SLSQP stopped
        assert obj[i] < g["objective"][i] + 1e-6


def test_line_search_statistics_of_the_c_oracle(c_oracle):
    """oracle_counters (developer counters of oracle/mpc_oracle.c): on 256 config-2 cold starts the filter line search takes fewer trial points per iteration than the l1 merit
    (measured 1.03 against 1.20) and refuses every trial step of a line search in well under 1 % of the iterations."""
    import ctypes as C
    from mpc_local_planner_amd import workloads as W
    lib = c_oracle._load()
    cfg = R.config_carlike_min_time(50)
    inp = W.carlike_min_time_inputs(256)
    per_iter = {}
    for ls in (0, 1):
        out = (C.c_longlong * 4)()
        lib.oracle_counters(out, 1)
        r = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg, line_search=ls), *inp)
        lib.oracle_counters(out, 1)
        iters, trials, refused, refused_trials = list(out)
        assert iters >= int(r[4].sum()) - 256 and trials >= iters and refused_trials <= trials      # (an iteration that ends a solve before its step is taken is not counted)
        per_iter[ls] = trials / iters
        assert refused <= 0.01 * iters
    assert per_iter[1] < per_iter[0] and per_iter[1] < 1.1

