"""Closed-loop scenarios for Controller::step: parameter sets, the stand-in "solver" that is plugged into BOTH the reference's Controller (oracle/ref_wrap_controller.cpp)
and the shipped facade (tests/host_harness/facade_step_host.cpp), and the event script (goal jumps, resets, state feedback, plans of 2..5 poses).
tests/golden/make_ref_vectors.py runs the REFERENCE through every scenario and records, per step, what it handed to the solver and what it returned
(tests/golden/ref_controller_steps.npz); tests/test_reference_pinned.py replays the recorded inputs on the facade."""
import copy

import numpy as np

from configure_cases import base_carlike

STEPS, NMAX = 40, 40


def variants():
    car = base_carlike()
    car["grid"]["variable_grid"]["grid_adaptation"]["max_grid_size"] = 36
    car["grid"]["variable_grid"]["grid_adaptation"]["min_grid_size"] = 3          # the batched solver needs 3 grid points (the reference allows 2: see the clamp test)
    v = {"carlike_example": car}
    a = copy.deepcopy(car); a["grid"]["variable_grid"]["enable"] = False; a["controller"]["outer_ocp_iterations"] = 1
    v["fixed_grid_warm_start_shifting"] = a
    a = copy.deepcopy(a); a["grid"]["warm_start"] = False
    v["fixed_grid_no_shifting"] = a
    a = copy.deepcopy(car); a["controller"]["prefer_x_feedback"] = True; a["controller"]["force_reinit_num_steps"] = 7; a["controller"]["outer_ocp_iterations"] = 2
    v["state_feedback_and_periodic_reinit"] = a
    a = copy.deepcopy(car); a["grid"]["xf_fixed"] = [True, True, False]; a["grid"]["variable_grid"]["grid_adaptation"]["enable"] = False
    v["partially_fixed_goal_no_adaptation"] = a
    a = copy.deepcopy(car); a["grid"]["variable_grid"]["grid_adaptation"]["max_grid_size"] = 24; a["grid"]["variable_grid"]["grid_adaptation"]["min_grid_size"] = 12
    a["controller"]["force_reinit_new_goal_dist"] = 0.04; a["controller"]["force_reinit_new_goal_angular"] = 0.05
    v["narrow_adaptation_range_touchy_reinit"] = a
    a = copy.deepcopy(car); a["robot"] = {"type": "unicycle"}
    a["planning"]["objective"] = {"type": "quadratic_form", "quadratic_form": {"state_weights": [2.0, 2.0, 0.25], "control_weights": [0.1, 0.05]}}
    a["grid"]["xf_fixed"] = [False, False, False]; a["grid"]["variable_grid"]["enable"] = False
    v["unicycle_quadratic_free_goal"] = a
    return v


def wrap(th):
    return (np.asarray(th, float) + np.pi) % (2.0 * np.pi) - np.pi


def stand_in_solver(x, u, dt, u_prev, dt_prev, state):
    """deterministic, not a solver: bends the states (not x_0, not the fixed goal components), mixes the controls with the previous control, scales dt so that the grid
    adaptation fires; `state`: calls (counter), dt_factor, free_dt, fixed (3 flags), fail_at (call numbers that report failure)"""
    n = x.shape[0]
    k = np.arange(n)[:, None]
    xs = x + 0.01 * np.sin(0.7 * k + state["calls"]) * np.array([1.0, -0.5, 0.2])
    xs[0] = x[0]
    fixed = np.asarray(state["fixed"], bool)
    xs[-1, fixed] = x[-1, fixed]
    xs[:, 2] = wrap(xs[:, 2])
    us = 0.9 * u + 0.05 * np.cos(np.arange(n - 1))[:, None] + 0.1 * u_prev + 0.01 * dt_prev
    dts = dt * state["dt_factor"] if state["free_dt"] else dt
    state["calls"] += 1
    return xs, us, dts, state["calls"] not in state["fail_at"]


def next_event(rng, pose, goal, t):
    """what happens before step: (goal', reset?, feedback (state, stamp) | None, dt_factor, plan)"""
    ev = int(rng.integers(0, 12))
    reset, fb = False, None
    if ev == 0:
        goal = goal + np.array([2.0, -1.0, 0.0])             # far away: the grid is re-initialised
    elif ev == 1:
        goal = goal + np.array([0.05, 0.02, 0.1])            # close by: the warm start goes on
    elif ev == 2:
        goal = goal.copy(); goal[2] = float(wrap(goal[2] + 2.0))
    elif ev == 3:
        reset = True
    elif ev == 4:
        fb = (pose + rng.normal(0, 0.05, 3), t - float(rng.choice([0.05, 0.5])))          # fresh (< 2 periods old) or stale
    factor = float(rng.choice([0.85, 1.0, 1.0, 1.2]))
    npl = int(rng.integers(2, 6))
    plan = np.linspace(pose, goal, npl)
    plan[1:-1, :2] += rng.normal(0, 0.1, (npl - 2, 2))
    plan[:, 2] = wrap(plan[:, 2])
    return goal, reset, fb, factor, plan
