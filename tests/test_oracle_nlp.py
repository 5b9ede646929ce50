"""CPU tests of the oracle's restatement of the reference NLP (oracle/se2_nlp.py).
The reference has no tests/golden vectors, so these pin the restatement to
hand-computed values of the reference formulas and to internal consistency."""
import math

import numpy as np
import pytest

from oracle import se2_nlp as R
from oracle import ipm_dense as I


def test_normalize_theta_interval_and_edges():
    # include/mpc_local_planner/utils/math_utils.h:81-91 : [-pi, pi)
    assert R.normalize_theta(0.3) == 0.3
    assert R.normalize_theta(math.pi) == pytest.approx(-math.pi)
    assert R.normalize_theta(-math.pi) == -math.pi
    assert R.normalize_theta(3 * math.pi + 0.1) == pytest.approx(-math.pi + 0.1)
    assert R.normalize_theta(-7.0) == pytest.approx(-7.0 + 2 * math.pi)
    v = R.normalize_theta(np.linspace(-20, 20, 401))
    assert np.all(v >= -math.pi) and np.all(v < math.pi)


def test_interpolate_angle_shortest_arc():
    # math_utils.h:100-103
    assert R.interpolate_angle(3.0, -3.0, 0.5) == pytest.approx(R.normalize_theta(3.0 + 0.5 * (2 * math.pi - 6.0)))
    assert R.interpolate_angle(0.0, 1.0, 2.0) == pytest.approx(2.0)


@pytest.mark.parametrize("model,p,expect", [
    (R.MODEL_UNICYCLE, (), lambda th, v, w: [v * math.cos(th), v * math.sin(th), w]),
    (R.MODEL_SIMPLE_CAR, (0.4,), lambda th, v, w: [v * math.cos(th), v * math.sin(th), v * math.tan(w) / 0.4]),
    (R.MODEL_SIMPLE_CAR_FRONT, (0.4,), lambda th, v, w: [v * math.cos(th), v * math.sin(th), v * math.sin(w) / 0.4]),
])
def test_dynamics_formulas(model, p, expect):
    x = np.array([0.3, -0.2, 0.7])
    u = np.array([0.35, -0.4])
    np.testing.assert_allclose(R.dynamics(model, p, x, u), expect(0.7, 0.35, -0.4), rtol=1e-15)


def test_bicycle_dynamics():
    # include/mpc_local_planner/systems/kinematic_bicycle_model.h:65-77
    lr, lf = 1.0, 1.5
    th, v, phi = 0.2, 0.3, 0.6
    beta = math.atan(lr / (lf + lr) * math.tan(phi))
    f = R.dynamics(R.MODEL_KINEMATIC_BICYCLE, (lr, lf), np.array([0, 0, th]), np.array([v, phi]))
    np.testing.assert_allclose(f, [v * math.cos(th + beta), v * math.sin(th + beta), v * math.sin(beta) / lr], rtol=1e-15)


def test_forward_collocation_wraps_heading_difference():
    # fd_collocation_se2.h:54-69: error = f - [(x2-x1)_xy, wrap(th2-th1)]/dt
    x1 = np.array([0.0, 0.0, 3.1])
    x2 = np.array([0.1, 0.2, -3.1])        # crosses +-pi: wrapped difference is 2*pi-6.2
    u = np.array([0.2, 0.1])
    e = R.collocation_defect(R.COLLOC_FORWARD, R.MODEL_UNICYCLE, (), x1, u, x2, 0.5)
    assert e[2] == pytest.approx(0.1 - (2 * math.pi - 6.2) / 0.5)
    assert e[0] == pytest.approx(0.2 * math.cos(3.1) - 0.1 / 0.5)


def test_midpoint_reduces_to_forward_and_crank_nicolson_follows_the_literal_reference_code():
    x1 = np.array([0.0, 0.0, 0.4]); x2 = np.array([0.1, 0.05, 0.4]); u = np.array([0.3, 0.0])
    f = R.collocation_defect(R.COLLOC_FORWARD, R.MODEL_UNICYCLE, (), x1, u, x2, 0.3)
    m = R.collocation_defect(R.COLLOC_MIDPOINT, R.MODEL_UNICYCLE, (), x1, u, x2, 0.3)
    c = R.collocation_defect(R.COLLOC_CRANK_NICOLSON, R.MODEL_UNICYCLE, (), x1, u, x2, 0.3)
    np.testing.assert_allclose(f, m, atol=1e-15)
    # fd_collocation_se2.h:139-141 evaluates  error -= quot - 0.5*(f1 + error)  with error == f2 on the
    # right-hand side, i.e. 1.5*f2 + 0.5*f1 - quot (NOT the textbook 0.5*(f1+f2) - quot); restated literally.
    fdyn = R.dynamics(R.MODEL_UNICYCLE, (), x1, u)
    quot = (x2 - x1) / 0.3
    np.testing.assert_allclose(c, 2.0 * fdyn - quot, atol=1e-15)


def test_cold_start_is_straight_line_with_shortest_arc_heading():
    # src/controller.cpp:807-857 + full_discretization_grid_base_se2.cpp:192-239 for a 2-pose plan
    cfg = R.config_carlike_min_time(11)
    x0 = np.array([0.0, 0.0, 3.0]); xf = np.array([1.0, 2.0, -3.0])
    t = R.cold_start(cfg, x0, xf)
    assert t.x.shape == (11, 3) and t.u.shape == (10, 2) and t.dt == 0.3
    np.testing.assert_allclose(t.x[5, :2], [0.5, 1.0])
    assert t.x[5, 2] == pytest.approx(R.normalize_theta(3.0 + 0.5 * (2 * math.pi - 6.0)))
    assert np.all(t.u == 0)
    np.testing.assert_array_equal(t.x[0], x0)
    np.testing.assert_array_equal(t.x[-1], xf)


def test_straight_line_init_heading_flip():
    # ...grid_base_se2.cpp:158-172: heading of the direction, flipped when the goal is behind
    cfg = R.config_carlike_min_time(5)
    t = R.initialize_sequences_straight_line(cfg, np.array([0, 0, 0.0]), np.array([-1.0, 0.0, 0.0]))
    assert t.x[1, 2] == pytest.approx(R.normalize_theta(math.pi + math.pi))
    assert t.x[0, 2] == 0.0


def test_variable_order_and_pack_roundtrip():
    # computeActiveVertices: u0, x1, u1, ..., x_{n-2}, u_{n-2}, [xf free], [dt]
    cfg = R.config_carlike_min_time(6)
    inp = R.CycleInputs(x0=np.zeros(3), xf=np.array([1, 0, 0.0]))
    nlp = R.ReferenceNlp(cfg, inp)
    assert nlp.nz == 2 + 5 * 4 + 0 + 1
    t = R.cold_start(cfg, inp.x0, inp.xf)
    t.u[:] = np.arange(10).reshape(5, 2)
    z = nlp.pack(t)
    np.testing.assert_array_equal(z[:2], t.u[0])
    np.testing.assert_array_equal(z[2:5], t.x[1])
    assert z[-1] == t.dt
    t2 = nlp.unpack(z)
    np.testing.assert_array_equal(t2.x, t.x)
    np.testing.assert_array_equal(t2.u, t.u)


def test_rate_rows_order_and_first_cycle_zeroing():
    # stage_inequality_se2.cpp:191-222: lower block first; k=0 with dt_prev==0 -> zeros (:197-201)
    cfg = R.config_carlike_min_time(4)
    inp = R.CycleInputs(x0=np.zeros(3), xf=np.array([1, 0, 0.0]), u_prev=np.array([0.1, 0.0]), dt_prev=0.0)
    nlp = R.ReferenceNlp(cfg, inp)
    t = R.cold_start(cfg, inp.x0, inp.xf)
    t.u[:] = [[0.2, 0.1], [0.3, 0.0], [0.1, -0.1]]
    g = nlp.inequalities(nlp.pack(t))
    assert g.size == 4 * 4
    np.testing.assert_array_equal(g[:4], 0.0)
    # k=1: (u1-u0)/dt = (0.1, -0.1)/0.3
    np.testing.assert_allclose(g[4:8], [-0.5 - 0.1 / 0.3, -0.5 + 0.1 / 0.3, 0.1 / 0.3 - 0.5, -0.1 / 0.3 - 0.5])
    # final rows against u_ref = 0: (0 - u2)/dt
    np.testing.assert_allclose(g[12:16], [-0.5 + 0.1 / 0.3, -0.5 - 0.1 / 0.3, -0.1 / 0.3 - 0.5, 0.1 / 0.3 - 0.5])


def test_solver_form_is_positive_rescaling_of_reference_form():
    cfg = R.config_carlike_min_time(9)
    rng = np.random.default_rng(1)
    inp = R.CycleInputs(x0=np.array([0, 0, 0.3]), xf=np.array([2.0, 1.0, 0.5]), u_prev=np.array([0.1, 0.05]), dt_prev=0.2)
    ref = R.ReferenceNlp(cfg, inp)
    snl = I.SolverNlp(cfg, inp)
    t = R.cold_start(cfg, inp.x0, inp.xf)
    v = snl.to_vec(t) + 0.05 * rng.standard_normal(snl.nv)
    tt = snl.to_traj(v)
    z = ref.pack(tt)
    ev = snl.eval(v)
    np.testing.assert_allclose(ev["c"], tt.dt * ref.equalities(z), atol=1e-14)
    assert ev["f"] == pytest.approx(ref.objective(z))
    gref = ref.inequalities(z)
    # reference rows: per stage k [lo0, lo1, hi0, hi1], final block last; solver rows identical order, scaled by dt_prev
    scale = np.concatenate([np.full(4, inp.dt_prev), np.full(gref.size - 4, tt.dt)])
    np.testing.assert_allclose(ev["g"], scale * gref, atol=1e-14)


@pytest.mark.parametrize("model,p", [(R.MODEL_UNICYCLE, ()), (R.MODEL_SIMPLE_CAR, (0.4,)), (R.MODEL_SIMPLE_CAR_FRONT, (0.4,)),
                                     (R.MODEL_KINEMATIC_BICYCLE, (1.0, 1.3))])
def test_model_derivatives_match_finite_differences(model, p):
    th, v, w = 0.4, 0.3, -0.5
    f, G, H = I.model_derivs(model, p, th, v, w)
    np.testing.assert_allclose(f, R.dynamics(model, p, np.array([0, 0, th]), np.array([v, w])), rtol=1e-14)
    q = np.array([th, v, w])
    h = 1e-6
    for j in range(3):
        e = np.zeros(3); e[j] = h
        fp, Gp, _ = I.model_derivs(model, p, *(q + e))
        fm, Gm, _ = I.model_derivs(model, p, *(q - e))
        np.testing.assert_allclose((fp - fm) / (2 * h), G[:, j], atol=1e-8)
        np.testing.assert_allclose((Gp - Gm) / (2 * h), H[:, :, j], atol=1e-7)


def test_solver_nlp_derivatives_vs_numeric_through_retraction():
    cfg = R.config_unicycle_quadratic(8)
    inp = R.CycleInputs(x0=np.array([0, 0, 0.1]), xf=np.array([1.0, 0.3, 0.2]), u_prev=np.zeros(2), dt_prev=0.2)
    nlp = I.SolverNlp(cfg, inp)
    rng = np.random.default_rng(3)
    v = nlp.to_vec(R.cold_start(cfg, inp.x0, inp.xf)) + 0.05 * rng.standard_normal(nlp.nv)
    lam = rng.standard_normal(nlp.mc)
    y = rng.uniform(0.1, 1, nlp.mg)
    ev = nlp.eval(v, lam, y, want_hess=True)

    def num(fun, h=1e-6):
        f0 = np.atleast_1d(fun(v))
        J = np.zeros((f0.size, v.size))
        for i in range(v.size):
            e = np.zeros(v.size); e[i] = h
            J[:, i] = (np.atleast_1d(fun(nlp.retract(v, e))) - np.atleast_1d(fun(nlp.retract(v, -e)))) / (2 * h)
        return J
    np.testing.assert_allclose(num(lambda a: nlp.eval(a)["c"]), ev["Jc"], atol=1e-7)
    np.testing.assert_allclose(num(lambda a: nlp.eval(a)["g"]), ev["Jg"], atol=1e-7)
    np.testing.assert_allclose(num(lambda a: nlp.eval(a)["f"])[0], ev["gf"], atol=1e-6)

    def gradL(a):
        e = nlp.eval(a)
        return e["gf"] + e["Jc"].T @ lam + e["Jg"].T @ y
    np.testing.assert_allclose(num(gradL), ev["W"], atol=1e-5)


FULL_Q = np.array([[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.4]])
FULL_R = np.array([[0.1, 0.02], [0.02, 0.05]])
FULL_QF = np.array([[8.0, 1.0, 0.0], [1.0, 9.0, 0.5], [0.0, 0.5, 0.6]])
FULL_S = np.array([[1.0, 0.2, 0.0], [0.2, 1.0, 0.1], [0.0, 0.1, 0.5]])


@pytest.mark.parametrize("case", ["full_weights", "trapezoid_free_dt", "trapezoid_fixed_dt", "hybrid", "full_weights_trapezoid_ball"])
def test_cost_variants_objective_equals_the_reference_form_and_derivatives_match(case):
    """full Q / R / Qf / S matrices (src/controller.cpp:561-592,652-668,686-702), trapezoidal rule for integral-form costs
    (finite_differences_grid_se2.cpp:63-68), hybrid minimum time + control cost (src/controller.cpp:616-618): the solver NLP's objective equals
    the reference-form objective at random points and its analytic derivatives match central differences through the retraction."""
    import dataclasses
    base = R.config_unicycle_quadratic(8)
    cfg = {
        "full_weights": dataclasses.replace(base, Q=FULL_Q, R=FULL_R, Qf=FULL_QF),
        "trapezoid_free_dt": dataclasses.replace(base, integral_form=True, cost_integration="trapezoidal_rule", dt_free=True),
        "trapezoid_fixed_dt": dataclasses.replace(base, integral_form=True, cost_integration="trapezoidal_rule"),
        "hybrid": dataclasses.replace(base, Q=np.zeros(3), Qf=None, hybrid_min_time=True, dt_free=True, xf_fixed=(True, True, True)),
        "full_weights_trapezoid_ball": dataclasses.replace(base, Q=FULL_Q, R=FULL_R, Qf=FULL_QF, integral_form=True, cost_integration="trapezoidal_rule", dt_free=True,
                                                           terminal_ball_S=FULL_S, terminal_ball_gamma=0.3),
    }[case]
    inp = R.CycleInputs(x0=np.array([0, 0, 0.1]), xf=np.array([1.0, 0.3, 0.2]), u_prev=np.zeros(2), dt_prev=0.2)
    nlp = I.SolverNlp(cfg, inp)
    ref = R.ReferenceNlp(cfg, inp)
    rng = np.random.default_rng(5)
    for trial in range(3):
        v = nlp.to_vec(R.cold_start(cfg, inp.x0, inp.xf)) + 0.05 * rng.standard_normal(nlp.nv)
        if nlp.idt >= 0:
            v[nlp.idt] = 0.3 + 0.1 * rng.uniform()
        tt = nlp.to_traj(v)
        assert abs(nlp.eval(v)["f"] - ref.objective(ref.pack(tt))) < 1e-12
    lam = rng.standard_normal(nlp.mc)
    y = rng.uniform(0.1, 1, nlp.mg)
    ev = nlp.eval(v, lam, y, want_hess=True)

    def num(fun, h=1e-6):
        f0 = np.atleast_1d(fun(v))
        J = np.zeros((f0.size, v.size))
        for i in range(v.size):
            e = np.zeros(v.size); e[i] = h
            J[:, i] = (np.atleast_1d(fun(nlp.retract(v, e))) - np.atleast_1d(fun(nlp.retract(v, -e)))) / (2 * h)
        return J
    np.testing.assert_allclose(num(lambda a: nlp.eval(a)["f"])[0], ev["gf"], atol=1e-6)
    np.testing.assert_allclose(num(lambda a: nlp.eval(a)["g"]), ev["Jg"], atol=1e-6)

    def gradL(a):
        e = nlp.eval(a)
        return e["gf"] + e["Jc"].T @ lam + e["Jg"].T @ y
    np.testing.assert_allclose(num(gradL), ev["W"], atol=2e-5)


def test_warm_start_shift_and_nearest_state():
    # ...grid_base_se2.cpp:241-339
    x = np.stack([np.linspace(0, 1, 6), np.zeros(6), np.zeros(6)], 1)
    u = np.arange(10, dtype=float).reshape(5, 2)
    t = R.Trajectory(x.copy(), u.copy(), 0.3)
    assert R.find_nearest_state(t, x[0]) == 0
    assert R.find_nearest_state(t, np.array([0.21, 0, 0])) == 1
    s = R.warm_start_shifting(t, np.array([0.21, 0, 0]))
    np.testing.assert_allclose(s.x[0], x[1])
    np.testing.assert_allclose(s.x[4], x[5])            # old xf moved into the sequence
    np.testing.assert_allclose(s.x[5], x[5] + (x[5] - x[4]))   # linear extrapolation of the tail
    np.testing.assert_allclose(s.u[0], u[1])
    np.testing.assert_allclose(s.u[4], s.u[3])          # last control held


def test_resample_keeps_endpoints_and_total_time():
    # ...grid_base_se2.cpp:440-524
    cfg = R.config_carlike_min_time(10)
    t = R.cold_start(cfg, np.array([0, 0, 0.0]), np.array([2.0, 1.0, 1.0]))
    for n_new in (9, 11):
        r = R.resample_trajectory(t, n_new)
        assert r.x.shape == (n_new, 3)
        np.testing.assert_allclose(r.x[0], t.x[0]); np.testing.assert_allclose(r.x[-1], t.x[-1])
        assert r.dt * (n_new - 1) == pytest.approx(t.dt * 9)
        # the straight line is reproduced exactly by linear re-interpolation
        np.testing.assert_allclose(r.x[:, 0], np.linspace(0, 2, n_new), atol=1e-12)
    a = R.adapt_grid_single_step(cfg, R.Trajectory(t.x, t.u, 0.5), n_max=50)
    assert a.x.shape[0] == 11
    b = R.adapt_grid_single_step(cfg, R.Trajectory(t.x, t.u, 0.2))
    assert b.x.shape[0] == 9


def test_time_series_output_duplicates_last_control():
    t = R.Trajectory(np.zeros((4, 3)), np.array([[1, 2], [3, 4], [5, 6.0]]), 0.25)
    tt, xs, us = R.time_series_output(t)
    np.testing.assert_allclose(tt, [0, 0.25, 0.5, 0.75])
    assert us.shape == (4, 2)
    np.testing.assert_array_equal(us[-1], us[-2])


def test_footprint_distances_and_obstacle_association():
    # teb semantics + src/optimal_control/stage_inequality_se2.cpp:50-162
    pt = R.Obstacle(R.OBST_POINT, np.array([[1.0, 0.0]]))
    assert R.footprint_distance(R.FOOTPRINT_POINT, (), np.array([0, 0, 0.0]), pt) == pytest.approx(1.0)
    assert R.footprint_distance(R.FOOTPRINT_CIRCLE, (0.3,), np.array([0, 0, 0.0]), pt) == pytest.approx(0.7)
    line_fp = (0.0, 0.0, 0.4, 0.0)
    assert R.footprint_distance(R.FOOTPRINT_LINE, line_fp, np.array([0, 0, 0.0]), pt) == pytest.approx(0.6)
    assert R.footprint_distance(R.FOOTPRINT_LINE, line_fp, np.array([0, 0, math.pi / 2]), pt) == pytest.approx(1.0)
    sq = R.Obstacle(R.OBST_POLYGON, np.array([[1, -1], [2, -1], [2, 1], [1, 1.0]]))
    assert R.footprint_distance(R.FOOTPRINT_POINT, (), np.array([0, 0, 0.0]), sq) == pytest.approx(1.0)
    # teb distance_point_to_polygon_2d has no inside test: a point inside gets its distance to the boundary
    assert R.footprint_distance(R.FOOTPRINT_POINT, (), np.array([1.5, 0, 0.0]), sq) == pytest.approx(0.5)
    np.testing.assert_allclose(sq.centroid(), [1.5, 0.0])
    # test node scenario (src/test_mpc_optim_node.cpp:67-69,105-106): 3 point obstacles
    cfg = R.OcpConfig(n=20, min_obstacle_dist=0.5, force_inclusion_dist=0.5, cutoff_dist=2.0)
    obst = [R.Obstacle(R.OBST_POINT, np.array([[-3.0, 1.0]])), R.Obstacle(R.OBST_POINT, np.array([[6.0, 2.0]])),
            R.Obstacle(R.OBST_POINT, np.array([[4.0, 0.1]]))]
    traj = R.cold_start(cfg, np.array([0, 0, 0.0]), np.array([5.0, 2.0, 0.0]))
    rel, rel_dyn = R.associate_obstacles(cfg, traj, obst)
    assert rel[0] == [] and all(len(r) <= 2 for r in rel)
    assert 0 not in sum(rel, [])                 # (-3,1) is beyond the cutoff of every pose
    assert any(2 in r for r in rel)
    assert all(len(r) == 0 for r in rel_dyn)


def test_footprint_distances_polygon_and_segment_cases():
    """teb distance semantics for the footprint / obstacle pairs the device does not take yet (reference-form restatement only):
    segment and polygon footprints against line and polygon obstacles -- closed edge loops, 0 where edges cross, no inside tests."""
    sq = R.Obstacle(R.OBST_POLYGON, np.array([[1, -1], [2, -1], [2, 1], [1, 1.0]]))
    ln = R.Obstacle(R.OBST_LINE, np.array([[0.0, 2.0], [3.0, 2.0]]))
    line_fp = (0.0, 0.0, 0.4, 0.0)
    assert R.footprint_distance(R.FOOTPRINT_LINE, line_fp, np.array([0, 0, 0.0]), sq) == pytest.approx(0.6)
    assert R.footprint_distance(R.FOOTPRINT_LINE, line_fp, np.array([0.8, 0, 0.0]), sq) == 0.0                 # crosses the left edge
    assert R.footprint_distance(R.FOOTPRINT_LINE, line_fp, np.array([1.3, 0, 0.0]), sq) == pytest.approx(0.3)  # fully inside: distance to the boundary
    assert R.footprint_distance(R.FOOTPRINT_LINE, line_fp, np.array([0, 0, math.pi / 2]), ln) == pytest.approx(1.6)
    tri = (0.0, 0.0, 0.5, 0.0, 0.0, 0.5)
    assert R.footprint_distance(R.FOOTPRINT_POLYGON, tri, np.array([0, 0, 0.0]), sq) == pytest.approx(0.5)
    assert R.footprint_distance(R.FOOTPRINT_POLYGON, tri, np.array([0, 0, 0.0]), ln) == pytest.approx(1.5)
    assert R.footprint_distance(R.FOOTPRINT_POLYGON, tri, np.array([0.7, 0, 0.0]), sq) == 0.0
    pt = R.Obstacle(R.OBST_POINT, np.array([[0.1, 0.1]]))
    assert R.footprint_distance(R.FOOTPRINT_POLYGON, tri, np.array([0, 0, 0.0]), pt) == pytest.approx(0.1)      # inside the footprint: boundary distance
