"""PINNED to reference code: the angle helpers (include/mpc_local_planner/utils/math_utils.h:36-91), the four robot models
(include/mpc_local_planner/systems/unicycle_robot.h, simple_car.h, kinematic_bicycle_model.h) and the three SE(2) collocation rules
(include/mpc_local_planner/optimal_control/fd_collocation_se2.h:45-153).

tests/golden/ref_models_collocation.npz holds what THOSE reference sources compute (compiled from /root/reference into oracle/_ref by
`make -C oracle ref`; Eigen / corbo / ROS / teb are replaced by the interface stand-ins of oracle/ref_stubs/, the arithmetic statements that run
are the reference's own -- oracle/ref_wrap.cpp).  Held to them here: the numpy oracle (reference-form NLP rows), the C oracle and the C++ facade
through the solver-form rows, and the HOST BUILD of the kernel's core (mpc_core.hpp, the code the HIP kernel runs).  Where /root/reference exists
the compiled reference code is also called directly on fresh inputs.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import se2_nlp as R
from oracle import ref_lib as RL

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_models_collocation.npz"))
MODELS = {0: (), 1: (0.4,), 2: (0.4,), 3: (1.0, 1.3)}


def test_angle_helpers_reproduce_the_reference_bit_for_bit():
    ours = np.array([R.normalize_theta(t) for t in G["theta"]])
    assert np.array_equal(ours, G["normalize_theta"])
    assert ((G["normalize_theta"] >= -np.pi) & (G["normalize_theta"] < np.pi)).all()          # [-pi, pi): +pi maps to -pi
    ours = np.array([R.interpolate_angle(a, b, f) for a, b, f in zip(G["a1"], G["a2"], G["factor"])])
    assert np.array_equal(ours, G["interpolate_angle"])
    from oracle import candidates as OC
    assert np.array_equal(OC.wrap(G["theta"]), G["normalize_theta"])


@pytest.mark.parametrize("model", sorted(MODELS))
def test_robot_models_reproduce_the_reference(model):
    ours = np.array([R.dynamics(model, MODELS[model], x, u) for x, u in zip(G["x1"], G["u"])])
    assert np.abs(ours - G[f"dynamics_model{model}"]).max() <= 1e-16


@pytest.mark.parametrize("model", sorted(MODELS))
@pytest.mark.parametrize("method", [0, 1, 2])
def test_reference_form_collocation_rows_reproduce_the_reference(model, method):
    """oracle/se2_nlp.py::collocation_defect IS the reference's computeEqualityConstraint, the literal Crank-Nicolson rule (1.5 f(x2) + 0.5 f(x1): `error`
    is aliased on its right-hand side, fd_collocation_se2.h:139-141) included"""
    ref = G[f"collocation_model{model}_method{method}"]
    ours = np.stack([R.collocation_defect(method, model, MODELS[model], G["x1"][i:i + 1], G["u"][i:i + 1], G["x2"][i:i + 1], G["dt"][i])[0] for i in range(ref.shape[0])])
    assert np.abs(ours - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max())


@pytest.fixture(scope="module")
def host():
    src = os.path.join(HERE, "host_harness", "host_solver.cpp")
    out = os.path.join(HERE, "host_harness", "_build", "libmpc_hostdbg_pin.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out], check=True)
    lib = C.CDLL(out)
    lib.hostdbg_normalize_theta.restype = C.c_double
    lib.hostdbg_normalize_theta.argtypes = [C.c_double]
    return lib


def test_kernel_core_angle_wrap_reproduces_the_reference(host):
    ours = np.array([host.hostdbg_normalize_theta(float(t)) for t in G["theta"]])
    assert np.array_equal(ours, G["normalize_theta"])


VX = np.load(os.path.join(HERE, "golden", "ref_vertex.npz"))


def test_vertex_retraction_reproduces_the_reference_s_vertex_classes(host):
    """SURVEY.md 8 row a15, pinned to EXECUTED reference code since r05: include/mpc_local_planner/optimal_control/vector_vertex_se2.h compiles as it is against the
    corbo vertex interface of oracle/ref_stubs/ (no stand-in shadows it any more) and its plus / plusUnfixed / setData / set were recorded on 400 vertices, 64 of them
    with the heading at +-pi, one ulp inside, and increments of 0, +-1e-17, +-1e-9, +-pi, 2 pi.  Held to the recording, bit for bit: the numpy oracle's retraction
    (se2_nlp.ReferenceNlp.plus / ipm_dense retract: z + dz, heading wrapped), the candidates' wrap, and the host build of the kernel core's accept step
    (x += alpha dx; heading = normalize_theta, mpc_wave.hpp::xt / accept) at alpha = 1."""
    v, d = VX["values"], VX["inc"]
    ours = v + d
    ours[:, 2] = [R.normalize_theta(t) for t in ours[:, 2]]
    assert np.array_equal(ours, VX["plus"]) and np.array_equal(VX["plus"], VX["plus_per_component"])
    assert ((VX["plus"][:, 2] >= -np.pi) & (VX["plus"][:, 2] < np.pi)).all()
    from oracle import candidates as OC
    assert np.array_equal(OC.wrap(v[:, 2] + d[:, 2]), VX["plus"][:, 2])
    # dimension 5: the tail behind the pose is plain reals
    o5 = VX["values5"] + VX["inc5"]; o5[:, 2] = [R.normalize_theta(t) for t in o5[:, 2]]
    assert np.array_equal(o5, VX["plus5"])
    # the kernel core's accept step
    out = np.empty_like(v)
    host.hostdbg_retract.restype = None
    host.hostdbg_retract(C.c_int(v.shape[0]), np.ascontiguousarray(v).ctypes.data_as(C.c_void_p), np.ascontiguousarray(d).ctypes.data_as(C.c_void_p), C.c_double(1.0),
                         out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, VX["plus"])
    # partially fixed vertex (the final state with fixed goal components): only the free components move, the increment carries one entry per free component
    for i in range(v.shape[0]):
        fx = VX["fixed"][i % 8].astype(bool)
        want = v[i].copy()
        want[~fx] += d[i][~fx]
        if not fx[2]:
            want[2] = R.normalize_theta(want[2])
        assert np.array_equal(want, VX["plus_unfixed"][i]) and VX["dim_unfixed"][i] == int((~fx).sum())
    # setData / set wrap the heading they are given
    raw = v[:64] + np.array([0.0, 0.0, 4.0])
    want = raw.copy(); want[:, 2] = [R.normalize_theta(t) for t in raw[:, 2]]
    assert np.array_equal(want, VX["set_data"]) and np.array_equal(want, VX["set_values"])
    # finite bounds: [lower, upper, any] over all components, then over the unfixed ones (what the NLP's bound multipliers are counted from)
    lb, ub = VX["bound_lb"], VX["bound_ub"]
    inf = 2e30
    for fx, got in zip(VX["fixed"].astype(bool), VX["bound_counts"]):
        allc = [int((lb > -inf).sum()), int((ub < inf).sum()), int(((lb > -inf) | (ub < inf)).sum())]
        free = [int(((lb > -inf) & ~fx).sum()), int(((ub < inf) & ~fx).sum()), int((((lb > -inf) | (ub < inf)) & ~fx).sum())] if (~fx).any() else allc
        assert list(got) == allc + free, (fx, got)


@pytest.mark.parametrize("model", sorted(MODELS))
@pytest.mark.parametrize("method", [0, 1, 2])
def test_kernel_core_collocation_rows_reproduce_the_reference_on_the_heading_manifold(host, model, method):
    """The kernel forms its rows as c = dt F(theta_1, u, dt) - (x_2 - x_1) with the midpoint / second Crank-Nicolson point placed by the explicit heading
    relation (mpc_core.hpp::model_trig_colloc) instead of interpolate_angle(theta_1, theta_2): the same thing wherever theta_2 satisfies the rule's own
    heading row -- so the comparison with the compiled reference rule is made there (theta_2 = theta_1 + dt f_2 for forward / midpoint differences,
    theta_1 + 2 dt f_2 for the literal Crank-Nicolson rule); positions are arbitrary.  dt * reference error == kernel row."""
    import mpc_local_planner_amd as m
    par = MODELS[model]
    x1, u, dt = G["x1"].copy(), G["u"].copy(), G["dt"].copy()
    f = G[f"dynamics_model{model}"]                                      # the heading rate does not depend on the pose
    x2 = G["x2"].copy()
    x2[:, 2] = G[f"manifold_theta2_model{model}_method{method}"]
    ref = G[f"manifold_collocation_model{model}_method{method}"] * dt[:, None]
    cfg = m.make_config(model=model, model_params=par if par else (0.0,), n=20, collocation=method)
    c = np.zeros_like(x1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    host.hostdbg_colloc(C.byref(cfg), C.c_int(x1.shape[0]), p(x1), p(u), p(x2), p(dt), p(c))
    assert np.abs(c - ref).max() < 5e-15, np.abs(c - ref).max()
    small = np.abs((2.0 if method == 2 else 1.0) * dt * f[:, 2]) < 3.0           # (a heading step beyond pi wraps: there the row is 2 pi on BOTH sides)
    assert np.abs(c[small, 2]).max() < 5e-15                              # on the heading manifold the heading row vanishes


@pytest.mark.skipif(RL.load() is None and not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_compiled_reference_code_directly_on_fresh_inputs():
    assert RL.build()
    rng = np.random.default_rng(99)
    th = rng.uniform(-50, 50, 500)
    assert np.array_equal(RL.normalize_theta(th), np.array([R.normalize_theta(t) for t in th]))
    sets = rng.uniform(-np.pi, np.pi, (20, 5))
    ours = np.array([np.arctan2(np.sin(r).sum(), np.cos(r).sum()) for r in sets])
    assert np.abs(np.array([RL.average_angles(r) for r in sets]) - ours).max() < 1e-15
    x1 = rng.uniform(-2, 2, (100, 3)); x2 = x1 + rng.uniform(-0.3, 0.3, (100, 3)); u = rng.uniform(-1, 1, (100, 2)); dt = rng.uniform(0.05, 0.4, 100)
    for model, par in MODELS.items():
        for method in (0, 1, 2):
            ref = RL.collocation(method, model, par, x1, u, x2, dt)
            ours = np.stack([R.collocation_defect(method, model, par, x1[i:i + 1], u[i:i + 1], x2[i:i + 1], dt[i])[0] for i in range(100)])
            assert np.abs(ours - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max())


# ---- StageInequalitySE2 (src/optimal_control/stage_inequality_se2.cpp, compiled from the reference): association, clearance rows, control-rate rows ---------
S = np.load(os.path.join(HERE, "golden", "ref_stage_inequality.npz"))


def _scene(i):
    import dataclasses
    n, no = int(S["n"][i]), int(S["n_obst"][i])
    dmin, fi, co, en, dtk = S["params"][i]
    cfg = dataclasses.replace(R.config_unicycle_quadratic(n), min_obstacle_dist=float(dmin), force_inclusion_dist=float(fi), cutoff_dist=float(co), enable_dynamic_obstacles=bool(en))
    obs = [R.Obstacle(R.OBST_POINT, S["obst_xy"][i, j:j + 1].copy(), 0.0, S["obst_vel"][i, j].copy() if S["dynamic"][i, j] else None) for j in range(no)]
    return cfg, R.Trajectory(S["states"][i, :n].copy(), np.zeros((n - 1, 2)), float(dtk)), obs, n


def test_obstacle_association_reproduces_the_reference():
    """StageInequalitySE2::update (:50-162) executed on 60 scenes of point obstacles (static and moving): every obstacle closer than force_inclusion_dist, the
    nearest one on the left and on the right inside cutoff_dist -- `cross2d(orientation, centroid)` with the centroid as an ABSOLUTE vector (:121) --, moving
    obstacles all kept when enabled; same obstacles in the same order at every grid point"""
    points = 0
    for i in range(S["n"].shape[0]):
        cfg, traj, obs, n = _scene(i)
        rel, rel_dyn = R.associate_obstacles(cfg, traj, obs, None)
        for k in range(n):
            assert rel[k] == S["rel"][i, k, :S["rel_cnt"][i, k]].tolist(), (i, k)
            assert rel_dyn[k] == S["dyn"][i, k, :S["dyn_cnt"][i, k]].tolist(), (i, k)
            points += 1
    assert points > 700 and S["rel_cnt"].max() >= 4 and S["dyn_cnt"].max() >= 2


def test_clearance_rows_reproduce_the_reference():
    """computeNonIntegralStateTerm / computeNonIntegralStateDtTerm (:164-189): min_obstacle_dist - distance, a moving obstacle predicted k * dt ahead"""
    worst = 0.0
    for i in range(S["n"].shape[0]):
        cfg, traj, obs, n = _scene(i)
        for k in range(1, n):
            for c, j in enumerate(S["rel"][i, k, :S["rel_cnt"][i, k]]):
                ours = cfg.min_obstacle_dist - R.footprint_distance(cfg.footprint_kind, cfg.footprint_params, traj.x[k], obs[j])
                worst = max(worst, abs(ours - S["rows"][i, k, c]))
            for c, j in enumerate(S["dyn"][i, k, :S["dyn_cnt"][i, k]]):
                ours = cfg.min_obstacle_dist - R.footprint_distance(cfg.footprint_kind, cfg.footprint_params, traj.x[k], obs[j], k * traj.dt)
                worst = max(worst, abs(ours - S["dyn_rows"][i, k, c]))
    assert worst < 1e-14


def test_control_rate_rows_reproduce_the_reference():
    """computeNonIntegralControlDeviationTerm (:191-226): finite lower bounds first, then finite upper bounds; all zero in the first cycle (k = 0, dt_prev = 0)"""
    import dataclasses
    inf = float(S["corbo_inf"])
    for i in range(S["rate_k"].shape[0]):
        lb = np.where(S["rate_lb"][i] <= -inf, -R.INF, S["rate_lb"][i]); ub = np.where(S["rate_ub"][i] >= inf, R.INF, S["rate_ub"][i])
        cfg = dataclasses.replace(R.config_unicycle_quadratic(5), du_lb=lb, du_ub=ub)
        inp = R.CycleInputs(x0=np.zeros(3), xf=np.ones(3), u_prev=S["rate_u_prev"][i], dt_prev=float(S["rate_dt_prev"][i]))
        nlp = R.ReferenceNlp(cfg, inp)
        m = int(S["rate_count"][i])
        assert m == len(nlp.du_lb_finite) + len(nlp.du_ub_finite)
        if S["rate_dt_prev"][i] == 0.0:
            assert (S["rate_rows"][i, :m] == 0).all()            # ReferenceNlp.inequalities writes zeros there as well (se2_nlp.py, k == 0 branch)
            continue
        ours = np.array(nlp._rate_rows(S["rate_u"][i], S["rate_u_prev"][i], float(S["rate_dt_prev"][i])))
        assert np.abs(ours - S["rate_rows"][i, :m]).max(initial=0.0) < 1e-15


def test_c_oracle_association_reproduces_the_reference(c_oracle):
    """the C oracle's obst_associate (the code the device mirrors line by line), with room for every row: moving obstacles first, then exactly the
    reference's static list in the reference's order"""
    for i in range(S["n"].shape[0]):
        cfg, traj, obs, n = _scene(i)
        no = len(obs)
        ob = c_oracle.obst_from_nlp_config(cfg, max_obstacles=max(no, 1), max_vertices=1, max_rows=32)
        oi, dropped = c_oracle.associate_at(c_oracle.from_nlp_config(cfg), ob, traj.x, S["obst_xy"][i, :no, None, :], np.ones(no, np.int32), velocity=S["obst_vel"][i, :no])
        assert dropped == 0
        for k in range(1, n):
            want = S["dyn"][i, k, :S["dyn_cnt"][i, k]].tolist() + S["rel"][i, k, :S["rel_cnt"][i, k]].tolist()
            assert [j for j in oi[k] if j >= 0] == want, (i, k)


# ---- MinTimeViaPointsCost (src/optimal_control/min_time_via_points_cost.cpp, compiled from the reference) ---------------------------------------------------
V = np.load(os.path.join(HERE, "golden", "ref_via_points.npz"))


def test_via_point_association_and_terms_reproduce_the_reference():
    """update() (:39-117): the closest grid state (first minimum; the final state only if strictly closer -- findClosestPose of the reference's own grid
    class, see the grid tests below), ordered mode restarting two states behind the previous match, a via-point at / beyond the goal moved to the state in front of it, one at /
    behind the start skipped (ordered: attached to state 1).  computeNonIntegralStateTerm (:130-145): position weight x squared distance plus -- as coded --
    the orientation weight times the wrapped heading difference, NOT squared.  computeNonIntegralDtTerm (:119-128): (n - 1) dt on the single-dt grid."""
    import dataclasses
    n_att = 0
    for i in range(V["n"].shape[0]):
        n, nv = int(V["n"][i]), int(V["n_via"][i])
        wp, wo, ordered, dtk = V["params"][i]
        x, via = V["states"][i, :n], V["via"][i, :nv]
        cfg = dataclasses.replace(R.config_carlike_min_time(n), objective=R.OBJ_MIN_TIME_VIA_POINTS, via_points_ordered=bool(ordered), vp_position_weight=float(wp), vp_orientation_weight=float(wo))
        ours = R.associate_via_points(cfg, x, via)
        assert ours == V["attached"][i, :nv].tolist(), i
        for v in range(nv):
            k = ours[v]
            if k < 0:
                continue
            n_att += 1
            term = wp * ((via[v, 0] - x[k, 0]) ** 2 + (via[v, 1] - x[k, 1]) ** 2) + (wo * R.normalize_theta(via[v, 2] - x[k, 2]) if wo > 0 else 0.0)
            assert abs(term - V["terms"][i, v]) < 1e-14
        assert abs(V["dt_term"][i] - (n - 1) * dtk) < 1e-15
        # the reference-form objective = time term + the attached via-points' terms
        inp = R.CycleInputs(x0=x[0], xf=x[-1], u_prev=np.zeros(2), dt_prev=0.0, via_points=via)
        nlp = R.ReferenceNlp(cfg, inp, via_idx=ours)
        J = nlp.objective(nlp.pack(R.Trajectory(x.copy(), np.zeros((n - 1, 2)), float(dtk))))
        assert abs(J - (V["dt_term"][i] + sum(V["terms"][i, v] for v in range(nv) if ours[v] >= 0))) < 1e-12
    assert n_att > 300 and (V["attached"] == -1).sum() > 5


# ---- the grid classes and TimeSeriesSE2 (src/optimal_control/full_discretization_grid_base_se2.cpp, finite_differences_variable_grid_se2.cpp,
# src/utils/time_series_se2.cpp), compiled and executed: oracle/ref_wrap_grid.cpp -> tests/golden/ref_grid.npz
GR = np.load(os.path.join(HERE, "golden", "ref_grid.npz"))


def _traj(i):
    n = int(GR["n"][i])
    return n, GR["x"][i, :n].copy(), GR["u"][i, :n - 1].copy(), float(GR["dt"][i])


def test_cold_start_reproduces_the_reference_grid():
    """update() on an empty grid (:58-134 -> initializeSequences, both overloads :136-239): the straight line with the heading of the direction of travel
    (reversed when the goal lies behind the robot, untouched start and goal poses), and the samples of an initial state trajectory between start and goal"""
    behind = 0
    for i in range(GR["n"].shape[0]):
        n = int(GR["n"][i])
        x0, xf, dt_ref = GR["cold_x0"][i], GR["cold_xf"][i], float(GR["cold_dt_ref"][i])
        cfg = R.OcpConfig(n=n, dt_ref=dt_ref)
        t = R.initialize_sequences_straight_line(cfg, x0, xf)
        assert np.array_equal(t.x, GR["cold_line_x"][i, :n]), i
        assert np.array_equal(t.u, GR["cold_line_u"][i, :n - 1]) and t.dt == dt_ref
        d = xf[:2] - x0[:2]
        behind += int(d @ np.array([np.cos(x0[2]), np.sin(x0[2])]) < 0)
        # xinit overload: the table is what DiscreteTimeReferenceTrajectory hands out at k dt_ref; equal time stamps = exact hits of the interpolation
        xi = GR["cold_xinit"][i, :n]
        t2 = R.initialize_sequences_xinit(cfg, x0, xf, np.arange(n) * dt_ref, xi)
        assert np.array_equal(t2.x, GR["cold_xinit_x"][i, :n]), i
    assert behind > 20


def test_time_series_se2_interpolation_reproduces_the_reference():
    """TimeSeriesSE2::getValuesInterpolate (src/utils/time_series_se2.cpp:34-111), Linear + ZeroOrderHold beyond the end: exact hits within the 1e-6 tolerance
    return the stored sample, the heading is interpolated on the circle"""
    for i in range(GR["ts_m"].shape[0]):
        m = int(GR["ts_m"][i])
        tm, vals = GR["ts_times"][i, :m], GR["ts_values"][i, :m]
        for q, ref, ref_nh, ok_nh in zip(GR["ts_query"][i], GR["ts_out"][i], GR["ts_out_no_hold"][i], GR["ts_ok_no_hold"][i]):
            ours = R.time_series_se2_interpolate(tm, vals, float(q))
            assert np.array_equal(ours, ref), (i, q)
            # NoExtrapolation: as coded (:43-46 `break` leaves the switch only) the call does NOT fail beyond the last stamp, it extends the last interval linearly
            assert ok_nh
            if q <= tm[-1]:
                assert np.array_equal(ref_nh, ref)
            elif q - tm[-1] < 1e-6:
                assert np.array_equal(ref_nh, vals[-1])
            else:
                fr = (q - tm[-2]) / (tm[-1] - tm[-2])
                lin = vals[-2] + fr * (vals[-1] - vals[-2]); lin[2] = R.interpolate_angle(vals[-2][2], vals[-1][2], fr)
                assert np.array_equal(ref_nh, lin)


def test_nearest_state_and_warm_start_shift_reproduce_the_reference_grid():
    """findNearestState (:304-339: stops at the first non-improving state, looks at most 20 states ahead, never at the final state) and the next cycle of the fixed
    grid (update(): warmStartShifting :241-302, then x_0 := measured state, fixed goal components := the goal :101-116)"""
    shifts = set()
    for i in range(GR["n"].shape[0]):
        n, x, u, dt = _traj(i)
        q, goal, fx = GR["query"][i], GR["goal_new"][i], GR["xf_fixed"][i]
        tr = R.Trajectory(x, u, dt)
        ns = R.find_nearest_state(tr, q)
        assert ns == GR["nearest"][i], i
        shifts.add(ns)
        cfg = R.OcpConfig(n=n, dt_ref=dt, xf_fixed=tuple(bool(f) for f in fx))
        w = R.new_run_overwrite(cfg, R.warm_start_shifting(tr, q), q, goal)
        assert np.array_equal(w.x, GR["warm_x"][i, :n]), i
        assert np.array_equal(w.u, GR["warm_u"][i, :n - 1]), i
    assert 0 in shifts and 20 in shifts and max(shifts) == 20 and len(shifts) > 12


def test_resampling_and_grid_adaptation_reproduce_the_reference_grid():
    """resampleTrajectory (:440-524) and adaptGridTimeBasedSingleStep (finite_differences_variable_grid_se2.cpp:99-121: one state more when dt > dt_ref (1 + hyst)
    and n < n_max, one less when dt < dt_ref (1 - hyst) and n > n_min)"""
    grown = shrunk = kept = 0
    for i in range(GR["n"].shape[0]):
        n, x, u, dt = _traj(i)
        tr = R.Trajectory(x, u, dt)
        n_new = int(GR["n_new"][i])
        r = R.resample_trajectory(tr, n_new)
        assert r.x.shape[0] == n_new
        assert np.array_equal(r.x, GR["resample_x"][i, :n_new]), i
        assert np.array_equal(r.u, GR["resample_u"][i, :n_new - 1]), i
        assert r.dt == GR["resample_dt"][i]
        dtr, n_max, n_min, hyst = GR["adapt_par"][i]
        a = R.adapt_grid_single_step(R.OcpConfig(n=n, dt_ref=float(dtr)), tr, n_min=int(n_min), n_max=int(n_max), hyst=float(hyst))
        na = int(GR["adapt_n"][i])
        assert a.x.shape[0] == na, i
        assert np.array_equal(a.x, GR["adapt_x"][i, :na]) and np.array_equal(a.u, GR["adapt_u"][i, :na - 1]) and a.dt == GR["adapt_dt"][i]
        grown += na > n; shrunk += na < n; kept += na == n
    assert grown > 15 and shrunk > 15 and kept > 30


def test_closest_pose_and_time_series_reproduce_the_reference_grid():
    """findClosestPose (:364-388) and getStateAndControlTimeSeries (:579-615: the last control repeated at the final state's time stamp)"""
    for i in range(GR["n"].shape[0]):
        n, x, u, dt = _traj(i)
        for (xr, yr, start), ref in zip(GR["closest_query"][i], GR["closest"][i]):
            assert R.find_closest_pose(x, float(xr), float(yr), int(start)) == ref, i
        t, xs, us = R.time_series_output(R.Trajectory(x, u, dt))
        assert np.array_equal(t, GR["series_t"][i, :n]) and np.array_equal(xs, GR["series_x"][i, :n]) and np.array_equal(us, GR["series_u"][i, :n])
        msg = R.optimal_control_result(xs, us, dt, True, 0.0, 0)
        assert np.array_equal(msg["states"], GR["series_x"][i, :n].reshape(-1)) and np.array_equal(msg["time_controls"], GR["series_t"][i, :n])


@pytest.fixture(scope="module")
def facade():
    src = os.path.join(HERE, "host_harness", "controller_host.cpp")
    out = os.path.join(HERE, "host_harness", "_build", "libctl_host_pinned.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wl,--unresolved-symbols=ignore-all", src, "-o", out], check=True)
    l = C.CDLL(out)
    l.ctl_resample.restype = C.c_double
    l.ctl_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int]
    l.ctl_interpolate_se2.restype = None
    l.ctl_interpolate_se2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    return l


def test_controller_facade_grid_logic_reproduces_the_reference_grid(facade):
    """the shipped host logic (include/mpc_controller.hpp: find_nearest_state, warm_start_shifting, resample_trajectory, interpolate_se2) against the recorded outputs
    of the reference's grid class; the facade keeps the controls as [n][2] with the last row repeated"""
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for i in range(GR["n"].shape[0]):
        n, x, u, dt = _traj(i)
        q = np.ascontiguousarray(GR["query"][i])
        xx = x.copy(); uu = np.ascontiguousarray(np.vstack([u, u[-1:]]))
        assert facade.ctl_find_nearest_state(p(xx), C.c_int(n), p(q)) == GR["nearest"][i]
        facade.ctl_warm_start_shifting(p(xx), p(uu), C.c_int(n), p(q))
        wx = GR["warm_x"][i, :n].copy()            # the recorded cycle also overwrote the start and the fixed goal components: compare what shifting alone decides
        assert np.array_equal(xx[1:n - 1], wx[1:n - 1]), i
        assert np.array_equal(uu[:n - 1], GR["warm_u"][i, :n - 1]), i
        free = GR["xf_fixed"][i] == 0
        assert np.array_equal(xx[n - 1][free], wx[n - 1][free])
        n_new = int(GR["n_new"][i]); cap = max(n, n_new)
        xb = np.zeros((cap, 3)); xb[:n] = x
        ub = np.zeros((cap, 2)); ub[:n - 1] = u; ub[n - 1] = u[-1]
        dt_new = facade.ctl_resample(p(xb), p(ub), dt, n, n_new)
        assert dt_new == GR["resample_dt"][i]
        assert np.array_equal(xb[:n_new], GR["resample_x"][i, :n_new]), i
        assert np.array_equal(ub[:n_new - 1], GR["resample_u"][i, :n_new - 1]), i
    for i in range(GR["ts_m"].shape[0]):
        m = int(GR["ts_m"][i])
        tm, vals = np.ascontiguousarray(GR["ts_times"][i, :m]), np.ascontiguousarray(GR["ts_values"][i, :m])
        out = np.zeros(3)
        for qv, ref in zip(GR["ts_query"][i], GR["ts_out"][i]):
            facade.ctl_interpolate_se2(m, p(tm), p(vals), float(qv), p(out))
            assert np.array_equal(out, ref), (i, qv)


# ---- the cost and terminal-condition classes (src/optimal_control/quadratic_cost_se2.cpp, final_state_conditions_se2.cpp), compiled and executed:
# oracle/ref_wrap_cost.cpp -> tests/golden/ref_costs.npz
CO = np.load(os.path.join(HERE, "golden", "ref_costs.npz"))


def _cost_scene(i, diagonal):
    n = int(CO["n"][i])
    pick = (lambda M: np.diag(M).copy()) if diagonal else (lambda M: M.copy())
    return n, CO["x"][i, :n], CO["u"][i, :n], CO["goal"][i], float(CO["dt"][i]), pick(CO["Q"][i]), pick(CO["R"][i]), pick(CO["Qf"][i]), pick(CO["S"][i]), float(CO["gamma"][i])


@pytest.mark.parametrize("diagonal", [False, True])
def test_cost_terms_reproduce_the_reference(diagonal):
    """per grid point: xd = x_k - x_ref with the heading difference wrapped (quadratic_cost_se2.cpp:36-37), then xd' Q xd (state term), xd' Q xd + u' R u (the
    integrand of the integral form, control reference zero or not), xd' Qf xd (final_state_conditions_se2.cpp:30-52), xd' S xd - gamma (:54-64)"""
    sfx = "_diag" if diagonal else ""
    wrapped = 0
    for i in range(CO["n"].shape[0]):
        n, x, u, goal, dt, Q, Rw, Qf, S, gamma = _cost_scene(i, diagonal)
        Qm, Rm, Qfm, Sm = (R.weight_matrix(w) for w in (Q, Rw, Qf, S))
        for k in range(n):
            xd = x[k] - goal
            wrapped += abs(xd[2]) > np.pi
            xd[2] = R.normalize_theta(xd[2])
            st = float(xd @ Qm @ xd)
            tol = 1e-14 * max(1.0, abs(st))
            assert abs(st - CO["form_state" + sfx][i, k]) < tol and abs(st - CO["state_state" + sfx][i, k]) < tol
            assert abs(st + float(u[k] @ Rm @ u[k]) - CO["form_l" + sfx][i, k]) < 4 * tol + 1e-14 * float(u[k] @ Rm @ u[k])
            assert abs(st - CO["state_l" + sfx][i, k]) < tol                                       # QuadraticStateCostSE2: no control part
            assert abs(float(xd @ Qfm @ xd) - CO["final" + sfx][i, k]) < 1e-14 * max(1.0, CO["final" + sfx][i, k])
            assert abs(float(xd @ Sm @ xd) - gamma - CO["ball" + sfx][i, k]) < 1e-14 * max(1.0, abs(CO["ball" + sfx][i, k]) + gamma)
            if not diagonal:
                ud = u[k] - CO["u_ref"][i]
                assert abs(st + float(ud @ Rm @ ud) - CO["form_l_uref"][i, k]) < 1e-13 * max(1.0, CO["form_l_uref"][i, k])
            else:
                # least-squares form (lsq solvers), diagonal weights: sqrt(Q) xd, three values per grid point whose squares sum to the quadratic form
                assert np.allclose(CO["form_state_lsq_diag"][i, k], np.sqrt(Q) * xd, rtol=1e-15, atol=1e-15)
                assert abs(float((CO["form_state_lsq_diag"][i, k] ** 2).sum()) - st) < 1e-13 * max(1.0, st)
                assert np.allclose(CO["final_lsq_diag"][i, k], np.sqrt(Qf) * xd, rtol=1e-15, atol=1e-15)
    assert wrapped > 200


@pytest.mark.parametrize("diagonal", [False, True])
@pytest.mark.parametrize("form", ["non_integral", "left_sum", "trapezoidal_rule"])
def test_reference_form_objective_is_the_sum_of_the_reference_s_terms(diagonal, form):
    """ReferenceNlp.objective against the recorded per-point terms put together the way the grid's edges do (finite_differences_grid_se2.cpp:53-126: one term per
    grid point k < n-1, or dt l(x_k, u_k) [left sum], or 0.5 dt (l(x_k, u_k) + l(x_{k+1}, u_k)) [trapezoidal rule], plus the final-state cost at x_{n-1}), and the
    terminal-ball row"""
    import dataclasses
    sfx = "_diag" if diagonal else ""
    for i in range(CO["n"].shape[0]):
        n, x, u, goal, dt, Q, Rw, Qf, S, gamma = _cost_scene(i, diagonal)
        Rm = R.weight_matrix(Rw)
        cfg = dataclasses.replace(R.config_unicycle_quadratic(n), Q=Q, R=Rw, Qf=Qf, terminal_ball_S=S, terminal_ball_gamma=gamma, xf_fixed=(False, False, False),
                                  integral_form=form != "non_integral", cost_integration="left_sum" if form == "non_integral" else form, dt_free=False, dt_ref=dt,
                                  du_lb=np.full(2, -R.INF), du_ub=np.full(2, R.INF))
        nlp = R.ReferenceNlp(cfg, R.CycleInputs(x0=x[0], xf=goal))
        z = nlp.pack(R.Trajectory(x.copy(), u[:n - 1].copy(), dt))
        J = nlp.objective(z)
        if form == "non_integral":
            ref = sum(CO["form_state" + sfx][i, k] + float(u[k] @ Rm @ u[k]) for k in range(n - 1))        # the control term is corbo's base class: u' R u
        elif form == "left_sum":
            ref = sum(dt * CO["form_l" + sfx][i, k] for k in range(n - 1))
        else:
            ref = sum(0.5 * dt * (CO["form_l" + sfx][i, k] + CO["form_l_next" + sfx][i, k]) for k in range(n - 1))
        ref += CO["final" + sfx][i, n - 1]
        assert abs(J - ref) < 1e-12 * max(1.0, abs(ref)), (i, J, ref)
        g = nlp.inequalities(z)
        assert abs(g[-1] - CO["ball" + sfx][i, n - 1]) < 1e-13 * max(1.0, abs(g[-1]))


# ---- Controller::step of the reference (src/controller.cpp:102-179, :807-857), executed in closed loop with a stand-in solver (oracle/ref_wrap_controller.cpp), against
# the shipped facade (include/mpc_controller.hpp) driven through the same script with the same stand-in (tests/host_harness/facade_step_host.cpp records what the
# facade hands to mpc_solve_batch).  Recorded: tests/golden/ref_controller_steps.npz (generator: make_ref_vectors.py + controller_scenarios.py)
import sys
sys.path.insert(0, os.path.join(HERE, "golden"))
import controller_scenarios as CS      # noqa: E402

STEPS_REC = np.load(os.path.join(HERE, "golden", "ref_controller_steps.npz"))
OPT_NAMES = ["grid_adaptation", "max_grid_size", "dt_hyst_ratio", "min_grid_size", "n_max", "warm_start", "outer_ocp_iterations", "force_reinit_new_goal_dist",
             "force_reinit_new_goal_angular", "allow_init_with_backward_motion", "force_reinit_num_steps", "prefer_x_feedback", "publish_ocp_results", "print_cpu_time"]
_SOLVER_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double)


@pytest.fixture(scope="module")
def facade_lib():
    src = os.path.join(HERE, "host_harness", "facade_step_host.cpp")
    out = os.path.join(HERE, "host_harness", "_build", "libfacade_step.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # -Bsymbolic: the recorder's own mpc_* definitions must win even when the product library was loaded globally earlier in the session
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic", src, "-o", out], check=True)
    lib = C.CDLL(out)
    V, D, I = C.c_void_p, C.c_double, C.c_int
    lib.fs_create.restype = V; lib.fs_create.argtypes = [V, V, I]
    lib.fs_destroy.argtypes = [V]; lib.fs_set_solver.argtypes = [V]; lib.fs_reset.argtypes = [V]
    lib.fs_set_previous_control.argtypes = [V, V, D]; lib.fs_state_feedback.argtypes = [V, V, D]
    lib.fs_step.restype = I; lib.fs_step.argtypes = [V, I, V, V, D, D, I, V, V, V, V]
    lib.fs_last_guess.restype = I; lib.fs_last_guess.argtypes = [I, V, V, V, V]
    lib.fs_result_msg.restype = None; lib.fs_result_msg.argtypes = [V, I] + [V] * 8
    return lib


class _Facade:
    def __init__(self, lib, params, solver, reference_reinit_sampling=True):
        from mpc_local_planner_amd import params as PP
        self.lib = lib
        self.cfg, ctrl, _ = PP.config_from_params(params)
        opt = np.array([float(ctrl.get(k, 0)) for k in OPT_NAMES])
        self.h = lib.fs_create(C.byref(self.cfg), opt.ctypes.data_as(C.c_void_p), 1)
        assert self.h

        def cb(n, px, pu, pdt, pup, dtp):
            x = np.ctypeslib.as_array(px, (n, 3)); u = np.ctypeslib.as_array(pu, (n - 1, 2))
            xs, us, dts, ok = solver(x.copy(), u.copy(), float(pdt[0]), np.array([pup[0], pup[1]]), float(dtp))
            x[:] = xs; u[:] = us; pdt[0] = dts
            return 1 if ok else 0
        self._cb = _SOLVER_CB(cb)

    def step(self, plan, u_prev, dt, t):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.fs_set_solver(C.cast(self._cb, C.c_void_p))
        up = np.ascontiguousarray(u_prev, float); self.lib.fs_set_previous_control(self.h, p(up), dt)
        plan = np.ascontiguousarray(plan, float); vel = np.array([0.1, 0.0, 0.05])
        to, xo, uo, n = np.zeros(256), np.zeros((256, 3)), np.zeros((256, 2)), C.c_int(0)
        ok = self.lib.fs_step(self.h, plan.shape[0], p(plan), p(vel), dt, t, 256, p(to), p(xo), p(uo), C.byref(n))
        gx, gu, gdt, cold = np.zeros((256, 3)), np.zeros((256, 2)), np.zeros(1), C.c_int(0)
        m = self.lib.fs_last_guess(256, p(gx), p(gu), p(gdt), C.byref(cold))
        return bool(ok), to[:n.value].copy(), xo[:n.value].copy(), uo[:n.value].copy(), gx[:m].copy(), gu[:m - 1].copy(), float(gdt[0])

    def close(self):
        self.lib.fs_destroy(self.h)


@pytest.mark.parametrize("variant", sorted(CS.variants()))
@pytest.mark.parametrize("script", [0, 1])
def test_controller_facade_steps_like_the_reference_s_controller(facade_lib, variant, script):
    """per control cycle: the state the cycle starts from (odometry pose, or a fresh state measurement with prefer_x_feedback), the re-initialisation decision (goal jump in
    distance / heading, reset(), force_reinit_num_steps), the initial state trajectory from plans of 2..5 poses (intermediate headings from the direction of travel) sampled
    as the reference samples it -- INCLUDING the stale-dt sampling of a re-initialisation after a first solve --, warm-start shifting on the fixed grid, single-step grid
    adaptation on every outer iteration, outer_ocp_iterations, the previous control handed over, and the time series given back: vertex values handed to the solver and
    results agree with the recorded run of the reference's Controller to 1e-12 at every one of 40 steps"""
    prm = CS.variants()[variant]
    key = lambda k: STEPS_REC[f"{variant}/{script}/{k}"]
    from mpc_local_planner_amd import params as PP
    cfg = PP.config_from_params(prm)[0]
    state = dict(calls=0, dt_factor=1.0, free_dt=bool(cfg.dt_free), fixed=[bool(f) for f in cfg.xf_fixed], fail_at={17, 60})
    fac = _Facade(facade_lib, prm, lambda *a: CS.stand_in_solver(*a, state))
    reinit_with_stale_dt = 0
    for i in range(CS.STEPS):
        if key("reset")[i]:
            facade_lib.fs_reset(fac.h)
        fb = key("fb")[i]
        if not np.isnan(fb[0]):
            s3 = np.ascontiguousarray(fb[:3]); facade_lib.fs_state_feedback(fac.h, s3.ctypes.data_as(C.c_void_p), float(fb[3]))
        state["dt_factor"] = float(key("factor")[i])
        plan = key("plan")[i, :int(key("n_plan")[i])]
        ok, to, xo, uo, gx, gu, gdt = fac.step(plan, key("u_prev")[i], 0.1, float(key("t")[i]))
        n, m = int(key("n_guess")[i]), int(key("n_out")[i])
        assert gx.shape[0] == n and xo.shape[0] == m, (i, gx.shape, n)
        assert np.abs(gx - key("guess_x")[i, :n]).max() < 1e-12 and np.abs(gu - key("guess_u")[i, :n - 1]).max() < 1e-12 and abs(gdt - key("guess_dt")[i]) < 1e-15, i
        assert ok == bool(key("ok")[i]), i
        assert np.abs(xo - key("out_x")[i, :m]).max() < 1e-12 and np.abs(uo - key("out_u")[i, :m]).max() < 1e-12 and np.abs(to - key("out_t")[i, :m]).max() < 1e-12, i
        reinit_with_stale_dt += int(abs(key("xinit_sample_dt")[i] - cfg.dt_ref) > 1e-6)
    assert state["calls"] >= CS.STEPS
    if cfg.dt_free:
        assert reinit_with_stale_dt > 0          # the script contains re-initialisations after a solve: the reference sampled its plan at the last optimised dt
    fac.close()


FZ = np.load(os.path.join(HERE, "golden", "ref_feasibility_and_result.npz"))


def test_feasibility_check_asks_about_the_same_poses_as_the_reference():
    """Controller::isPoseTrajectoryFeasible (src/controller.cpp:859-917), executed with a recording costmap model: every grid point up to look_ahead_idx (out-of-range:
    all), between neighbours farther apart than the inscribed radius or turning more than min_resolution_collision_check_angular the accumulated intermediate poses;
    only an answer of -1 ends the check (-2 / -3 do not).  oracle/feasibility.py asks about the same poses in the same order and stops at the same call."""
    from oracle import feasibility as FE
    interpolated = 0
    for i in range(FZ["n"].shape[0]):
        n = int(FZ["n"][i]); x = FZ["x"][i, :n]
        r_in, ang, look = FZ["par"][i]
        asked = []
        ok = FE.is_pose_trajectory_feasible(None, 0.0, None, x, None, float(r_in), float(ang), int(look), pose_cost=lambda a, b, c: asked.append((a, b, c)) or 0.0)
        m = int(FZ["n_calls"][i])
        assert ok == bool(FZ["feasible"][i]) and len(asked) == m, (i, len(asked), m)
        assert np.abs(np.array(asked) - FZ["calls"][i, :m]).max() < 1e-13
        interpolated += m - (n if look < 0 or look >= n else int(look) + 1)
        hit, cnt = int(FZ["hit_at"][i]), [0]

        def blocked(a, b, c):
            cnt[0] += 1
            return -1.0 if cnt[0] - 1 == hit else (-2.0 if cnt[0] % 5 == 0 else 3.0)
        assert FE.is_pose_trajectory_feasible(None, 0.0, None, x, None, float(r_in), float(ang), int(look), pose_cost=blocked) == bool(FZ["hit_feasible"][i]) is False
        assert cnt[0] == FZ["hit_calls"][i] == hit + 1
    assert interpolated > 200


def test_result_message_of_the_facade_is_the_one_the_reference_publishes(facade_lib):
    """mpc_local_planner_msgs/OptimalControlResult as Controller::publishOptimalControlResult fills it (src/controller.cpp:197-221), taken from the stand-in publisher:
    header.seq counts from 0 (published before ++_ocp_seq), dim_states / dim_controls, the time series sample after sample, optimal_solution_found"""
    import copy
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import configure_cases
    prm = configure_cases.base_carlike()
    prm["controller"]["publish_ocp_results"] = True; prm["controller"]["outer_ocp_iterations"] = 1
    prm["grid"]["variable_grid"]["grid_adaptation"]["enable"] = False
    n = int(FZ["n"][0]); traj = FZ["x"][0, :n]
    prm["grid"]["grid_size_ref"] = n

    def put(x, u, dt, up, dtp):
        xs = traj.copy(); xs[0] = x[0]
        return xs, u + 0.1, 0.2, True
    fac = _Facade(facade_lib, prm, put)
    plan = np.stack([traj[0], traj[-1]])
    for step, seq in ((0, FZ["msg_seq"][0]), (1, FZ["msg_seq_second_step"][0])):
        ok, to, xo, uo, *_ = fac.step(plan, np.zeros(2), 0.1, 0.1 * step)
        head = np.zeros(9); a = [np.zeros(4 * n) for _ in range(4)]
        p = lambda v: v.ctypes.data_as(C.c_void_p)
        facade_lib.fs_result_msg(fac.h, xo.shape[0], p(to), p(np.ascontiguousarray(xo)), p(np.ascontiguousarray(uo)), p(head), *[p(v) for v in a])
        assert head[0] == seq == step and head[1] == FZ["msg_dim_states"][0] == 3 and head[2] == FZ["msg_dim_controls"][0] == 2 and head[3] == FZ["msg_optimal_solution_found"][0] == 1
        if step == 0:
            for got, name in zip(a, ("time_states", "states", "time_controls", "controls")):
                ref = FZ["msg_" + name]
                assert int(head[5 + ("time_states", "states", "time_controls", "controls").index(name)]) == ref.size
                assert np.abs(got[:ref.size] - ref).max() < 1e-12, name
    fac.close()


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_recorded_vectors_are_what_the_compiled_reference_gives_today():
    """the committed fixtures against a fresh run of oracle/_ref (built from /root/reference by this test): guards against fixtures that drift from the generator"""
    import json
    assert RL.build()
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import configure_cases
    rec = json.load(open(os.path.join(HERE, "golden", "ref_configure.json")))
    for name, params in configure_cases.cases().items():
        status, log = RL.probe_configure(params)
        assert status == rec[name]["status"] and [t for lv, t in log if lv == 3] == rec[name]["errors"], name
        if status == 1:
            ctl = RL.RefController(params)
            assert ctl.dump() == rec[name]["built"], name
            ctl.close()
    assert np.array_equal(RL.vertex_plus(VX["values"], VX["inc"]), VX["plus"]) and np.array_equal(RL.vertex_plus(VX["values5"], VX["inc5"]), VX["plus5"])
    for i in range(0, 400, 37):
        fx = VX["fixed"][i % 8]
        o, nu = RL.vertex_plus_unfixed(VX["values"][i], fx, VX["inc"][i][fx == 0])
        assert np.array_equal(o, VX["plus_unfixed"][i]) and nu == VX["dim_unfixed"][i]
    for i in range(0, GR["n"].shape[0], 7):
        n, x, u, dt = _traj(i)
        assert RL.grid_find_nearest_state(x, u, dt, GR["query"][i]) == GR["nearest"][i]
        wx, wu = RL.grid_warm_start_cycle(x, u, dt, GR["query"][i], GR["goal_new"][i], GR["xf_fixed"][i])
        assert np.array_equal(wx, GR["warm_x"][i, :n]) and np.array_equal(wu, GR["warm_u"][i, :n - 1])
        rx, ru, rdt = RL.grid_resample(x, u, dt, int(GR["n_new"][i]))
        assert np.array_equal(rx, GR["resample_x"][i, :rx.shape[0]]) and rdt == GR["resample_dt"][i]
    for i in range(0, CO["n"].shape[0], 9):
        n, x, u, goal, dt, Q, Rw, Qf, S_, gamma = _cost_scene(i, False)
        assert np.array_equal(RL.quadratic_cost(Q, Rw, x, goal, u, form=True, integral=True), CO["form_l"][i, :n])
        assert np.array_equal(RL.terminal_ball(S_, gamma, x, goal), CO["ball"][i, :n])


# ---- the reference's plugin source (src/mpc_local_planner_ros.cpp), compiled as a whole and executed for the functions that prepare the solver's inputs
# (oracle/ref_wrap_plugin.cpp -> tests/golden/ref_plugin_inputs.npz, ref_footprint_models.json)
PLG = np.load(os.path.join(HERE, "golden", "ref_plugin_inputs.npz"))


def test_costmap_scan_reproduces_the_reference_plugin():
    """updateObstacleContainerWithCostmap (:474-499): LETHAL cells only, the last row and column never visited, column-major visiting order, cell centres, the
    behind-the-robot filter; include_costmap_obstacles = false: nothing.  oracle/costmap.py (the CPU restatement the device kernel mpc_costmap_to_obstacles is compared
    with bit for bit in tests/test_gpu_parity.py) gives the same obstacles in the same order"""
    from oracle import costmap as CM
    total = 0
    for i in range(12):
        cost, par = PLG[f"cm{i}_cost"], PLG[f"cm{i}_par"]
        ours = CM.costmap_to_obstacles(cost, par[0], par[1:3], par[3:6], par[6])
        ref = PLG[f"cm{i}_obstacles"]
        assert ours.shape == ref.shape and (ref.size == 0 or np.array_equal(ours, ref)), i
        total += ref.shape[0]
        assert ref.shape[0] < int((cost[:-1, :-1] == 254).sum()) + 1
    assert total > 200 and PLG["cm_disabled"].shape[0] == 0


def test_plan_helpers_reproduce_the_reference_plugin(host_ctl):
    """updateViaPointsContainer (:619-635) and estimateLocalGoalOrientation (:807-852): the package's plugin_inputs.py and the C++ helpers of include/mpc_controller.hpp"""
    from mpc_local_planner_amd import plugin_inputs as PI
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    some = 0
    for i in range(PLG["vp_n"].shape[0]):
        n = int(PLG["vp_n"][i]); plan = np.ascontiguousarray(PLG["vp_plan"][i, :n]); sep = float(PLG["vp_sep"][i]); m = int(PLG["vp_count"][i])
        ref = PLG["vp_out"][i, :m]
        ours = PI.via_points_from_plan(plan, sep)
        assert ours.shape[0] == m and (m == 0 or (np.array_equal(ours[:, :2], ref[:, :2]) and np.abs(ours[:, 2] - ref[:, 2]).max() < 1e-15)), i      # yaw went through a quaternion there
        out = np.zeros((n + 1, 3))
        assert host_ctl.ctl_via_points_from_plan(n, p(plan), sep, p(out)) == m and np.array_equal(out[:m], ours)
        some += m
        idx, ma, yaw, tx, ty, gx, gy, gth = PLG["go_par"][i]
        goal, tr = np.array([gx, gy, gth]), np.array([yaw, tx, ty])
        a = PI.estimate_local_goal_orientation(plan, goal, int(idx), tr, int(ma))
        b = host_ctl.ctl_goal_orientation(n, p(plan), p(goal), int(idx), p(tr), int(ma))
        for got in (a, b):
            assert abs(np.arctan2(np.sin(got - PLG["go_out"][i]), np.cos(got - PLG["go_out"][i]))) < 1e-12, i
    assert some > 100


def test_obstacle_messages_reproduce_the_reference_plugin(host_ctl):
    """updateObstacleContainerWithCostmapConverter (:501-541) / updateObstacleContainerWithCustomObstacles (:543-617): kinds by the number of points and the radius, the
    planar transform of custom obstacles, the 1 mm/s threshold below which an obstacle stays static, the velocity going to the LAST obstacle of the container,
    messages without points.  plugin_inputs.py and the C++ ObstacleSet (through its mpc_obstacles view)"""
    from mpc_local_planner_amd import plugin_inputs as PI
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    kinds = set()
    for i in range(PLG["ms_n"].shape[0]):
        k = int(PLG["ms_n"][i]); conv = bool(PLG["ms_converter"][i]); tr = np.ascontiguousarray(PLG["ms_transform"][i])
        msgs = [{"points": [tuple(q) for q in PLG["ms_pts"][i, j, :int(PLG["ms_npts"][i, j])]], "radius": float(PLG["ms_radius"][i, j]), "velocity": tuple(PLG["ms_vel"][i, j])} for j in range(k)]
        ours = PI.obstacles_from_messages(msgs, conv, tr)
        m = int(PLG["ms_count"][i])
        assert len(ours) == m, i
        rec = np.zeros((8, 5)); verts = np.zeros((8, 6, 2))
        npts = np.ascontiguousarray(PLG["ms_npts"][i]); pts = np.ascontiguousarray(PLG["ms_pts"][i]); rad = np.ascontiguousarray(PLG["ms_radius"][i]); vel = np.ascontiguousarray(PLG["ms_vel"][i])
        assert host_ctl.ctl_obstacles_from_messages(k, 6, p(npts), p(pts), p(rad), p(vel), int(conv), p(tr), 8, p(rec), p(verts)) == m
        for o, (v_, r_, vel_) in enumerate(ours):
            kind, nv, radius, dyn, vx, vy = PLG["ms_rec"][i, o]
            kinds.add(int(kind))
            assert v_.shape[0] == nv == rec[o, 0] and r_ == radius == rec[o, 1]
            assert (kind == 1) == (nv == 1 and radius > 0) and (kind == 0) == (nv == 1 and radius == 0) and (kind == 2) == (nv == 2) and (kind == 3) == (nv > 2)
            assert np.abs(v_ - PLG["ms_verts"][i, o, :int(nv)]).max() < 1e-13 and np.abs(verts[o, :int(nv)] - PLG["ms_verts"][i, o, :int(nv)]).max() < 1e-13
            assert bool(dyn) == bool(np.any(vel_ != 0)) == bool(rec[o, 2]) and np.array_equal(vel_ * bool(dyn), [vx * dyn, vy * dyn]) and np.array_equal(rec[o, 3:5], vel_)
    assert kinds == {0, 1, 2, 3}
    # pack_obstacles: the arrays of struct mpc_obstacles; capacity overflow is an error, not a silent drop
    nobs, nv, vv, rr, vel = PI.pack_obstacles(ours, 8, 6)
    assert nobs == len(ours) and all(nv[o] == ours[o][0].shape[0] for o in range(nobs))
    with pytest.raises(ValueError):
        PI.pack_obstacles(ours + ours + ours + ours + ours + ours + ours + ours + ours, 8, 6)


@pytest.fixture(scope="module")
def host_ctl():
    src = os.path.join(HERE, "host_harness", "controller_host.cpp")
    out = os.path.join(HERE, "host_harness", "_build", "libctl_host_plugin_helpers.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wl,--unresolved-symbols=ignore-all", src, "-o", out], check=True)
    l = C.CDLL(out)
    V, D, I = C.c_void_p, C.c_double, C.c_int
    l.ctl_via_points_from_plan.restype = I; l.ctl_via_points_from_plan.argtypes = [I, V, D, V]
    l.ctl_goal_orientation.restype = D; l.ctl_goal_orientation.argtypes = [I, V, V, I, V, I]
    l.ctl_obstacles_from_messages.restype = I; l.ctl_obstacles_from_messages.argtypes = [I, I, V, V, V, V, I, V, I, V, V]
    l.ctl_prune_plan.restype = I; l.ctl_prune_plan.argtypes = [I, V, V, D, V]
    l.ctl_transform_plan.restype = I; l.ctl_transform_plan.argtypes = [I, V, V, I, I, D, D, V, V]
    return l


# ---- drop-in at the plugin level: the reference's plugin source src/mpc_local_planner_ros.cpp, UNCHANGED, built twice -- on the reference's own Controller
# (oracle/ref_wrap_plugin.cpp) and on include/mpc_reference_binding.hpp (oracle/ref_wrap_plugin_on_binding.cpp: the binding takes the place of the reference's controller.h,
# src/controller.cpp is not compiled, the C ABI is the recorder of tests/host_harness/facade_step_host.cpp) -- and run side by side
def _plugin_variants():
    import copy
    import configure_cases
    car = configure_cases.base_carlike()
    car["controller"]["outer_ocp_iterations"] = 1
    car["footprint_model"] = {"type": "line", "line_start": [0.0, 0.0], "line_end": [0.4, 0.0]}
    v = {"carlike_line_footprint": car}
    a = copy.deepcopy(car); a["controller"]["global_plan_viapoint_sep"] = 0.6; a["planning"]["objective"] = {"type": "minimum_time_via_points", "minimum_time_via_points": {"position_weight": 8.0}}
    a["controller"]["outer_ocp_iterations"] = 2; a["footprint_model"] = {"type": "polygon", "vertices": [[0.3, 0.2], [-0.3, 0.2], [-0.3, -0.2], [0.3, -0.2]]}
    v["via_points_polygon_footprint"] = a
    a = copy.deepcopy(car); a["robot"] = {"type": "unicycle"}; a["grid"]["variable_grid"]["enable"] = False; a["grid"]["xf_fixed"] = [False, False, False]
    a["planning"]["objective"] = {"type": "quadratic_form", "quadratic_form": {"state_weights": [2.0, 2.0, 0.25], "control_weights": [0.1, 0.05]}}
    a["controller"]["global_plan_overwrite_orientation"] = False; a["collision_avoidance"]["include_costmap_obstacles"] = False; a["footprint_model"] = {"type": "circular", "radius": 0.25}
    v["unicycle_quadratic_fixed_grid_no_costmap_obstacles"] = a
    a = copy.deepcopy(car); a["controller"]["prefer_x_feedback"] = True; a["controller"]["force_reinit_num_steps"] = 9
    v["state_feedback_preferred_periodic_reinit"] = a
    a = copy.deepcopy(car); a["controller"]["max_global_plan_lookahead_dist"] = 3.0; a["controller"]["global_plan_prune_distance"] = 0.5; a["collision_avoidance"]["costmap_obstacles_behind_robot_dist"] = 0.3
    a["collision_avoidance"]["collision_check_no_poses"] = 5; a["collision_avoidance"]["collision_check_min_resolution_angular"] = 0.2; a["footprint_model"] = {"type": "point"}
    v["long_lookahead_short_feasibility_check"] = a
    return v


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("variant", ["carlike_line_footprint", "via_points_polygon_footprint", "unicycle_quadratic_fixed_grid_no_costmap_obstacles", "long_lookahead_short_feasibility_check",
                                     "state_feedback_preferred_periodic_reinit"])
def test_reference_plugin_source_runs_unchanged_on_the_binding(variant):
    """initialize() -> setPlan() -> 40 x computeVelocityCommands() on a costmap with obstacles (a wall appears in front of the robot for four cycles: the feasibility check
    trips, the planner resets) and one failing solve: the two builds agree at every cycle in the mbf outcome code, the velocity command, the obstacle and via-point containers,
    the goal flag, the infeasible-plan counter, the planned trajectory and the vertex values handed to the solver"""
    assert RL.build()
    from mpc_local_planner_amd import params as PP
    prm = _plugin_variants()[variant]
    cfg = PP.config_from_params(prm)[0]
    rng = np.random.default_rng(7)
    cost = np.zeros((100, 120), np.uint8)
    for _ in range(25):
        i, j = rng.integers(5, 95), rng.integers(30, 115)
        cost[i:i + 2, j:j + 2] = 254
    fp = [(0.2, 0.1), (-0.2, 0.1), (-0.2, -0.1), (0.2, -0.1)]
    mk = lambda: dict(calls=0, dt_factor=0.97, free_dt=bool(cfg.dt_free), fixed=[bool(f) for f in cfg.xf_fixed], fail_at={9})
    sa, sb = mk(), mk()
    A = RL.PluginRunner(prm, cost, 0.1, (-2.0, -5.0), footprint=fp, solver=lambda *a: CS.stand_in_solver(*a, sa))
    B = RL.PluginRunner(prm, cost, 0.1, (-2.0, -5.0), footprint=fp, solver=lambda *a: CS.stand_in_solver(*a, sb), lib=RL.load_plugin_on_binding(), prefix="amd_plugin_")
    assert A.initialized and B.initialized
    plan = np.stack([np.linspace(0, 8, 60), 1.5 * np.sin(np.linspace(0, 3, 60)), np.zeros(60)], 1)
    plan[:-1, 2] = np.arctan2(np.diff(plan[:, 1]), np.diff(plan[:, 0])); plan[-1, 2] = plan[-2, 2]
    assert A.set_plan(plan) and B.set_plan(plan)
    pose = np.array([0.0, 0.0, 0.2])
    codes, checked = [], 0
    for i in range(40):
        if i == 20:
            k = int((pose[0] + 0.25 + 2.0) / 0.1); cost[:, k:k + 2] = 254
        if i == 24:
            cost[:, :] = 0
        if variant == "state_feedback_preferred_periodic_reinit" and i % 3 == 0:        # a measured state: fresh every sixth cycle, stale every sixth
            meas, stamp = pose + np.array([0.02, -0.01, 0.03]), (-0.05 if i % 6 == 0 else -1.0)
            A.state_feedback(meas, stamp); B.state_feedback(meas, stamp)
            last_meas, last_fresh = meas, stamp > -0.2                        # the stand-in clock stands at 0: a measurement stays fresh until the next message
        a, b = A.cycle(pose, (0.1, 0.0, 0.02), cost), B.cycle(pose, (0.1, 0.0, 0.02), cost)
        codes.append(a["code"])
        if variant == "state_feedback_preferred_periodic_reinit" and a["guess_x"].size:
            assert np.allclose(a["guess_x"][0], last_meas if last_fresh else pose, atol=1e-12), i          # the solve starts from the measured state only while it is fresh
        assert a["code"] == b["code"] and np.abs(a["cmd"] - b["cmd"]).max() < 1e-12, (i, a["code"], b["code"], a["cmd"], b["cmd"])
        assert (a["n_obstacles"], a["n_via"], a["goal_reached"], a["infeasible_in_a_row"]) == (b["n_obstacles"], b["n_via"], b["goal_reached"], b["infeasible_in_a_row"]), i
        assert a["x_seq"].shape == b["x_seq"].shape and (a["x_seq"].size == 0 or np.abs(a["x_seq"] - b["x_seq"]).max() < 1e-12), i
        assert a["feasibility_calls"] == b["feasibility_calls"] and abs(a["feasibility_checksum"] - b["feasibility_checksum"]) < 1e-9, i      # the poses the costmap model was asked about
        checked += a["feasibility_calls"]
        assert a["guess_x"].shape == b["guess_x"].shape and (a["guess_x"].size == 0 or (np.abs(a["guess_x"] - b["guess_x"]).max() < 1e-12 and abs(a["guess_dt"] - b["guess_dt"]) < 1e-15)), i
        if a["code"] == 0 and len(a["x_seq"]) > 1:
            pose = a["x_seq"][1].copy()
    assert checked > 200
    assert codes.count(0) >= 30 and codes.count(100) >= 2                     # SUCCESS and NO_VALID_CMD (failed solve; infeasible trajectory unless the check is short-sighted)
    if variant == "via_points_polygon_footprint":
        assert a["n_via"] > 0
    A.close(); B.close()


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_binding_hands_the_plugins_obstacle_container_to_the_abi():
    """costmap cells and custom obstacle messages (circle, point, line, polygon; moving ones) end up in the plugin's ObstContainer; the binding turns that container into the
    arrays of struct mpc_obstacles every cycle: same obstacles, same order, radius and velocity kept.  With a handle that holds fewer obstacles than the container, the
    nearest to the robot are kept (and a warning is logged)"""
    assert RL.build()
    import configure_cases
    prm = configure_cases.base_carlike()
    prm["controller"]["outer_ocp_iterations"] = 1; prm["collision_avoidance"]["enable_dynamic_obstacles"] = True
    prm["mpc_hip"] = {"max_obstacles": 40, "max_vertices": 6}
    cost = np.zeros((60, 80), np.uint8); cost[30:33, 40:43] = 254; cost[10:12, 60:62] = 254
    msgs = [{"points": [(1, 1, 0)], "radius": 0.3, "velocity": (0.1, 0.0)}, {"points": [(2, -1, 0)]}, {"points": [(1, 2, 0), (2, 2, 0)], "velocity": (0.0005, 0)},
            {"points": [(3, 1, 0), (3.5, 1, 0), (3.5, 1.5, 0), (3, 1.6, 0)], "velocity": (0, -0.2)}]
    for cap in (40, 8):
        prm["mpc_hip"]["max_obstacles"] = cap
        run = RL.PluginRunner(prm, cost, 0.1, (-2.0, -3.0), footprint=[(0.2, 0.1), (-0.2, 0.1), (-0.2, -0.1)], lib=RL.load_plugin_on_binding(), prefix="amd_plugin_")
        assert run.initialized and run.set_plan(np.linspace([0, 0, 0], [4, 0.5, 0], 30))
        run.set_custom_obstacles(msgs)
        pose = np.array([0.0, 0.0, 0.1])
        assert run.cycle(pose)["code"] == 0
        n_c, cont = run.container(); n_a, abi = run.abi_obstacles()
        assert n_c == 17 and len(cont) == 17                                   # 13 lethal cells inside the scan (9 + 4) + 4 messages
        if cap >= n_c:
            assert n_a == n_c
            for (v1, r1, vel1), (v2, r2, vel2) in zip(cont, abi):
                assert np.array_equal(v1, v2) and r1 == r2 and np.array_equal(vel1, vel2)
            assert [v.shape[0] for v, _, _ in abi[-4:]] == [1, 1, 2, 4] and abi[-4][1] == 0.3 and np.array_equal(abi[-4][2], [0.1, 0.0]) and np.array_equal(abi[-2][2], [0.0, 0.0])
        else:
            assert n_a == cap
            dist = np.array([np.hypot(*(v.mean(0) - pose[:2])) for v, _, _ in cont])          # centroid of points / segments / these polygons' vertices
            kept = np.sort(dist)[:cap]
            got = np.array([np.hypot(*(v.mean(0) - pose[:2])) for v, _, _ in abi])
            assert np.allclose(np.sort(got), kept, atol=1e-9)
        run.close()


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("rule", ["non_integral", "left_sum", "trapezoidal_rule"])
@pytest.mark.parametrize("xf_fixed", [(True, True, True), (False, False, False)])
def test_reference_form_nlp_is_the_reference_s_terms_on_the_reference_s_edge_layout(rule, xf_fixed):
    """the reference's createEdges (src/optimal_control/finite_differences_grid_se2.cpp:36-154), compiled and executed with record edges (oracle/ref_wrap_edges.cpp), says which
    vertices every term is built on: the previous control and ITS dt at k = 0, u_{k-1} and the grid's dt afterwards, xf as the successor of x_{n-2}, (u_ref, u_{n-2}, dt) for the
    closing control-deviation edge, final-state edges only while xf is not fixed.  Evaluating the EXECUTED reference terms (collocation rows, rate rows, cost integrands, final
    cost, terminal ball) on exactly those vertices reproduces oracle/se2_nlp.py::ReferenceNlp -- equalities, rate rows, terminal ball and objective"""
    import dataclasses
    assert RL.build()
    rng = np.random.default_rng(5)
    n = 7
    x = np.cumsum(rng.uniform(-0.1, 0.4, (n, 3)), 0); x[:, 2] = rng.uniform(-np.pi, np.pi, n)
    u = rng.normal(0, 0.3, (n - 1, 2)); dt = 0.23
    goal = x[-1] + rng.normal(0, 0.3, 3); goal[2] = float(R.normalize_theta(goal[2]))
    if all(xf_fixed):
        x[-1] = goal                                   # a fixed final state sits on the goal
    u_prev, dt_prev = rng.normal(0, 0.2, 2), 0.11
    Q, Rw, Qf, S, gamma = np.array([2.0, 1.5, 0.3]), np.array([0.4, 0.2]), np.array([5.0, 4.0, 1.0]), np.array([1.0, 1.0, 0.2]), 0.5
    du_lb, du_ub = np.array([-0.5, -0.6]), np.array([0.7, 0.8])
    cfg = dataclasses.replace(R.config_unicycle_quadratic(n), Q=Q, R=Rw, Qf=Qf, terminal_ball_S=S, terminal_ball_gamma=gamma, xf_fixed=xf_fixed, integral_form=rule != "non_integral",
                              cost_integration="left_sum" if rule == "non_integral" else rule, dt_free=False, dt_ref=dt, du_lb=du_lb, du_ub=du_ub, collocation=R.COLLOC_FORWARD)
    nlp = R.ReferenceNlp(cfg, R.CycleInputs(x0=x[0], xf=goal, u_prev=u_prev, dt_prev=dt_prev))
    z = nlp.pack(R.Trajectory(x.copy(), u.copy(), dt))
    edges = RL.create_edges(x, u, dt, [int(f) for f in xf_fixed], "trapezoidal_rule" if rule == "trapezoidal_rule" else "left_sum", cost_integral=rule != "non_integral",
                            final_cost=True, final_constraint="inequality")
    val = {f"x{k}": x[k] for k in range(n - 1)}
    val.update({f"u{k}": u[k] for k in range(n - 1)})
    val.update({"xf": x[-1], "dt": dt, "u_prev": u_prev, "u_prev_dt": dt_prev, "u_ref": np.zeros(2)})
    model_par = MODELS[cfg.model]
    J, eq, ineq = 0.0, [], []
    Qm, Rm = np.diag(Q), np.diag(Rw)
    for s_, kind, k, v in edges:
        if kind == "non_integral_stage_functions":
            xk, uk, dtk, up, dtp = (val[name] for name in v)
            assert (v[3], v[4]) == (("u_prev", "u_prev_dt") if k == 0 else (f"u{k - 1}", "dt")) and v[2] == "dt"
            if rule == "non_integral":                               # state term (executed reference) + corbo's control term u' R u
                J += RL.quadratic_cost(Qm, Rm, xk[None], goal, uk[None], form=True)[0] + float(uk @ Rm @ uk)
            ineq += list(RL.control_deviation_rows(k, uk, up, dtp, du_lb, du_ub))
        elif kind == "LeftSumCostEdge":
            xk, uk, dtk = (val[name] for name in v)
            J += dtk * RL.quadratic_cost(Qm, Rm, xk[None], goal, uk[None], form=True, integral=True)[0]
        elif kind == "TrapezoidalIntegralCostEdge":
            xk, uk, xn, dtk = (val[name] for name in v)
            assert v[2] == ("xf" if k == n - 2 else f"x{k + 1}")
            l = RL.quadratic_cost(Qm, Rm, np.stack([xk, xn]), goal, np.stack([uk, uk]), form=True, integral=True)
            J += 0.5 * dtk * (l[0] + l[1])
        elif kind == "FDCollocationEdge":
            xk, uk, xn, dtk = (val[name] for name in v)
            eq += list(RL.collocation(R.COLLOC_FORWARD, cfg.model, model_par, xk[None], uk[None], xn[None], dtk)[0])
        elif kind == "final_state_cost":
            J += RL.final_state_cost(np.diag(Qf), val[v[0]][None], goal)[0]
        elif kind == "final_state_constraint":
            ball = RL.terminal_ball(np.diag(S), gamma, val[v[0]][None], goal)[0]
        elif kind == "final_control_deviation":
            assert v == ["u_ref", f"u{n - 2}", "dt"]
            closing = list(RL.control_deviation_rows(k, val[v[0]], val[v[1]], val[v[2]], du_lb, du_ub))
    kinds = [e[1] for e in edges]
    assert ("final_state_cost" in kinds) == ("final_state_constraint" in kinds) == (not all(xf_fixed))
    if all(xf_fixed):
        ours_ineq = np.array(ineq + closing)
    else:
        ours_ineq = np.array(ineq + [ball] + closing)                 # the oracle lists: rate rows per grid point, the terminal ball, the closing rate rows
    assert np.abs(np.array(eq) - nlp.equalities(z)).max() < 1e-14
    g = nlp.inequalities(z)
    assert g.shape == ours_ineq.shape and np.abs(g - ours_ineq).max() < 1e-13
    assert abs(J - nlp.objective(z)) < 1e-12 * max(1.0, abs(J))


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("loop", ["carlike_line_footprint", "via_points_polygon_footprint", "diff_drive_quadratic_form", "carlike_to_the_goal", "carlike_block_close_to_the_path"])
def test_binding_reproduces_the_recorded_closed_loops_with_real_solves(loop):
    """tests/golden/ref_plugin_closed_loop_<loop>.npz was recorded with the reference's plugin on the reference's own Controller and the C oracle's solve behind it.  The same
    plugin source on the binding (recording C ABI, the same C oracle behind it), replaying the recorded poses: identical outcome codes, commands and planned trajectories to
    1e-9 at each of the 60 cycles -- the CPU twin of tests/test_gpu_reference_plugin.py::test_plugin_on_the_gpu_solver_reproduces_the_plugin_on_the_cpu_oracle"""
    import json
    assert RL.build()
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import plugin_oracle_solver
    from mpc_local_planner_amd import params as PP
    rec = np.load(os.path.join(HERE, "golden", f"ref_plugin_closed_loop_{loop}.npz"))
    prm = json.load(open(os.path.join(HERE, "golden", f"ref_plugin_closed_loop_{loop}.json")))
    res, ox, oy = rec["par"]
    run = RL.PluginRunner(prm, rec["cost"], float(res), (float(ox), float(oy)), footprint=rec["footprint"], lib=RL.load_plugin_on_binding(), prefix="amd_plugin_")
    run.solver = plugin_oracle_solver.make(run, PP.config_from_params(prm)[0])
    assert run.initialized and run.set_plan(rec["plan"])
    for i in range(rec["pose"].shape[0]):
        o = run.cycle(rec["pose"][i], rec["vel"][i])
        m = int(rec["n"][i])
        assert o["code"] == rec["code"][i] and o["x_seq"].shape[0] == m and o["n_via"] == rec["n_via"][i], i
        assert np.abs(o["cmd"] - rec["cmd"][i]).max() < 1e-9 and np.abs(o["x_seq"] - rec["x_seq"][i, :m]).max() < 1e-9, i
    run.close()


def test_plan_pruning_and_selection_reproduce_the_reference_plugin(host_ctl):
    """pruneGlobalPlan (:645-685) and transformGlobalPlan (:687-805), executed with a planar tf answer: what is cut off behind the robot (and that the function reports success
    even when no pose is close enough), where the selected part starts (closest pose inside 85 % of the costmap's half size), where it ends (radius, max_plan_length), the
    goal index, the poses moved into the planning frame; an empty selection yields the global goal"""
    from mpc_local_planner_amd import plugin_inputs as PI
    cut = injected = 0
    for i in range(PLG["gp_n"].shape[0]):
        n = int(PLG["gp_n"][i]); plan = PLG["gp_plan"][i, :n]
        yaw, tx, ty, px, py, pth, d, sx, sy, res, ml = PLG["gp_par"][i]
        ok, pr = PI.prune_global_plan(plan, (px, py, pth), (yaw, tx, ty), d)
        assert ok == bool(PLG["gp_pruned_ok"][i]) and pr.shape[0] == PLG["gp_pruned_n"][i] and (pr.shape[0] == 0 or np.array_equal(pr[0, :2], PLG["gp_pruned_first"][i])), i
        cut += pr.shape[0] < n
        tp, gi = PI.transform_global_plan(plan, (px, py, pth), int(sx), int(sy), res, ml, (yaw, tx, ty))
        m = int(PLG["gp_tr_n"][i]); ref = PLG["gp_tr"][i, :m]
        assert tp.shape[0] == m and gi == PLG["gp_goal_idx"][i], (i, tp.shape, m, gi)
        assert np.abs(tp[:, :2] - ref[:, :2]).max() < 1e-12 and np.abs(np.arctan2(np.sin(tp[:, 2] - ref[:, 2]), np.cos(tp[:, 2] - ref[:, 2]))).max() < 1e-12
        injected += int(m == 1 and gi == n - 1)
        if yaw == 0 and tx == 0 and ty == 0:                       # the C++ helpers of include/mpc_controller.hpp take a plan that is already in the planning frame
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            pl = np.ascontiguousarray(plan); rb = np.array([px, py, pth]); out = np.zeros((n + 1, 3)); mm = C.c_int(0)
            k = host_ctl.ctl_prune_plan(n, p(pl), p(rb), d, p(out))
            assert k == pr.shape[0] and np.array_equal(out[:k], pr)
            gj = host_ctl.ctl_transform_plan(n, p(pl), p(rb), int(sx), int(sy), res, ml, p(out), C.byref(mm))
            assert gj == gi and mm.value == m and np.array_equal(out[:m, :2], plan[gi - m + 1:gi + 1, :2] if not (m == 1 and gi == n - 1) else plan[-1:, :2])
    assert cut > 20 and injected >= 1


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_recorded_plugin_loop_is_what_the_reference_build_gives_today():
    """the generator's run of tests/golden/ref_plugin_closed_loop_carlike_line_footprint.npz, repeated live on the reference's plugin + the reference's Controller with the C oracle
    behind it: bit-identical commands and trajectories (guards the committed recording against drift of the generator, the stand-ins or the oracle)"""
    import json
    assert RL.build()
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import plugin_oracle_solver
    from mpc_local_planner_amd import params as PP
    rec = np.load(os.path.join(HERE, "golden", "ref_plugin_closed_loop_carlike_line_footprint.npz"))
    prm = json.load(open(os.path.join(HERE, "golden", "ref_plugin_closed_loop_carlike_line_footprint.json")))
    res, ox, oy = rec["par"]
    run = RL.PluginRunner(prm, rec["cost"], float(res), (float(ox), float(oy)), footprint=rec["footprint"])
    run.solver = plugin_oracle_solver.make(run, PP.config_from_params(prm)[0])
    assert run.initialized and run.set_plan(rec["plan"])
    for i in range(rec["pose"].shape[0]):
        o = run.cycle(rec["pose"][i], rec["vel"][i])
        m = int(rec["n"][i])
        assert o["code"] == rec["code"][i] and np.array_equal(o["cmd"], rec["cmd"][i]) and np.array_equal(o["x_seq"], rec["x_seq"][i, :m]), i
    run.close()


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_binding_configure_builds_the_handle_the_parameter_readers_describe():
    """mpc_local_planner::Controller of include/mpc_reference_binding.hpp, configure() on a ros::NodeHandle: the mpc_config it hands to mpc_create equals what the parameter
    reader gives for the same parameters plus the handle capacities (mpc_hip/*); footprint_model/type costmap_2d becomes the polygon of setCostmapFootprint, or -- without
    that call -- the point model with a warning; a parameter set the reference rejects makes configure() return false with the reference's message"""
    assert RL.build()
    import configure_cases
    from mpc_local_planner_amd import params as PP, _abi as A2
    lib = RL.load_plugin_on_binding()
    lib.amd_binding_configure.restype = C.c_int; lib.amd_binding_configure.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    lib.fs_last_created_config.restype = None; lib.fs_last_created_config.argtypes = [C.POINTER(A2.MpcConfig)]
    lib.amd_plugin_log.restype = C.c_int; lib.amd_plugin_log.argtypes = [C.c_char_p, C.c_int]

    def configure(tree, fp=None):
        a = None if fp is None else np.ascontiguousarray(fp, float)
        ok = lib.amd_binding_configure("\n".join(RL.plugin_param_lines(tree)).encode(), 0 if a is None else a.shape[0], None if a is None else a.ctypes.data_as(C.c_void_p))
        cfg = A2.MpcConfig(); lib.fs_last_created_config(C.byref(cfg))
        buf = C.create_string_buffer(1 << 14); lib.amd_plugin_log(buf, len(buf))
        return bool(ok), cfg, buf.value.decode()
    prm = configure_cases.base_carlike()
    prm["mpc_hip"] = {"max_obstacles": 48, "max_vertices": 5, "max_obstacle_rows": 6}
    prm["footprint_model"] = {"type": "line", "line_start": [0.0, 0.0], "line_end": [0.4, 0.0]}
    ok, cfg, _ = configure(prm)
    ref = PP.config_from_params(prm, max_obstacles=48, max_vertices=5, max_obstacle_rows=6, max_via_points=16)[0]
    assert ok and (cfg.max_obstacles, cfg.max_vertices, cfg.max_obstacle_rows) == (48, 5, 6) and cfg.n == 50         # sized for the largest grid of the adaptation
    for f in ("model", "dt_ref", "dt_free", "dt_lb", "dt_ub", "collocation", "objective", "max_iter", "tol", "hessian_mode", "min_obstacle_dist", "force_inclusion_dist", "cutoff_dist",
              "footprint_kind", "enable_dynamic_obstacles"):
        assert getattr(cfg, f) == getattr(ref, f), f
    for f in ("model_params", "xf_fixed", "u_lb", "u_ub", "du_lb", "du_ub", "footprint_params"):
        assert list(getattr(cfg, f)) == list(getattr(ref, f)), f
    square = [(0.2, 0.1), (-0.2, 0.1), (-0.2, -0.1), (0.2, -0.1)]
    prm["footprint_model"] = {"type": "costmap_2d"}
    ok, cfg, log = configure(prm, square)
    assert ok and cfg.footprint_kind == 4 and cfg.footprint_n_vertices == 4 and np.allclose(np.array(list(cfg.footprint_vertices)[:8]).reshape(4, 2), square)
    ok, cfg, log = configure(prm)
    assert ok and cfg.footprint_kind == 0 and "setCostmapFootprint" in log
    ok, _, log = configure(configure_cases.cases()["unknown_objective"])
    assert not ok and "Unknown objective type 'maximum_comfort'" in log


@pytest.mark.skipif(not os.path.isdir(RL.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_to_the_goal_loop_has_no_failing_cycle_with_the_acceptable_level_stop():
    """The reference's plugin + Controller, driven in closed loop towards the goal with the C solver behind it (tests/golden/ref_plugin_closed_loop_carlike_to_the_goal.npz).
    With Ipopt's acceptable-level stop -- which the reference's wrapper counts as success and which is the default of every solver in this repository -- every cycle answers
    SUCCESS and the goal is reached in cycle 52: the recording.  With the rule switched off (`acceptable_tol: 0` in the numeric options) the 4-point grid 0.27 m in front of the
    goal stalls just above tol and cycles from 49 on answer NO_VALID_CMD (what round 2 recorded for cycle 49)."""
    import json
    import copy
    assert RL.build()
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import plugin_oracle_solver
    from mpc_local_planner_amd import params as PP
    rec = np.load(os.path.join(HERE, "golden", "ref_plugin_closed_loop_carlike_to_the_goal.npz"))
    prm = json.load(open(os.path.join(HERE, "golden", "ref_plugin_closed_loop_carlike_to_the_goal.json")))
    res, ox, oy = rec["par"]

    def drive(prm):
        run = RL.PluginRunner(prm, rec["cost"], float(res), (float(ox), float(oy)), footprint=rec["footprint"])
        run.solver = plugin_oracle_solver.make(run, PP.config_from_params(prm)[0])
        assert run.initialized and run.set_plan(rec["plan"])
        pose, vel, codes, reached = np.array([0.0, 0.0, 0.1]), np.zeros(3), [], []
        for _ in range(rec["pose"].shape[0]):
            o = run.cycle(pose, vel)
            codes.append(int(o["code"])); reached.append(int(o["goal_reached"]))
            v, w = o["cmd"][0], o["cmd"][2]
            pose = pose + 0.1 * np.array([v * np.cos(pose[2]), v * np.sin(pose[2]), v / 0.4 * np.tan(w)])
            vel = np.array([v, 0.0, w])
        run.close()
        return np.array(codes), np.array(reached)
    codes, reached = drive(prm)
    assert np.array_equal(codes, rec["code"]) and np.array_equal(reached, rec["goal_reached"]) and not codes.any() and int(np.argmax(reached)) == 52
    off = copy.deepcopy(prm)
    off.setdefault("solver", {}).setdefault("ipopt", {}).setdefault("ipopt_numeric_options", {})["acceptable_tol"] = 0.0
    codes, reached2 = drive(off)
    bad = list(np.nonzero(codes)[0])
    assert bad and min(bad) >= 45, bad           # only the tiny grids right in front of the goal fail (round 3, monotone barrier rule: cycle 49 alone)
