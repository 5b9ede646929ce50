"""Candidate initial trajectories for one planner instance, HOST-SIDE variant: the guesses go in through x_init / u_init / dt_init as extra
members of the same batch and the winner is picked on the host.  Since round 2 the library does this on the device (mpc_config.n_candidates:
seeds generated in the kernel, hedged workgroups, deterministic lowest-converged-index rule) -- this module stays for
callers that bring their own guesses or want another selection rule (here: shortest transition time among the converged candidates).

BASELINE.json's north star names "candidate initial trajectories" next to independent planner instances as what a batch holds.  The
interior-point solve is local: which driving-direction reversals the solution contains is decided by the initial guess (DESIGN.md section 9,
"solver tolerance and Hessian"), so several guesses per instance raise the converged fraction and pick better local solutions.  Host-side only:
the guesses go in through x_init / u_init / dt_init of the C ABI.

Guesses (vertex values; controls zero = the solver seeds them from the states, dt = dt_ref):
  cold_start_guess        what Controller::step builds from a 2-pose plan (src/controller.cpp:807-857 + initializeSequences with xinit,
                          src/optimal_control/full_discretization_grid_base_se2.cpp:192-239): straight line, heading on the shortest arc
  travel_direction_guess  initializeSequences without xinit (...grid_base_se2.cpp:136-190): straight line, heading = direction of travel,
                          turned by pi when the goal lies behind the start pose; reverse=True takes the other driving direction (ours)
"""
from __future__ import annotations

import numpy as np


def _wrap(th):
    """normalize_theta (include/mpc_local_planner/utils/math_utils.h:81-91), vectorised"""
    th = np.asarray(th, float)
    out = th - np.floor(th / (2 * np.pi)) * 2 * np.pi
    out = np.where(out >= np.pi, out - 2 * np.pi, out)
    out = np.where(out < -np.pi, out + 2 * np.pi, out)
    return np.where((th >= -np.pi) & (th < np.pi), th, out)


def cold_start_guess(x0, xf, n: int, dt_ref: float):
    x0 = np.asarray(x0, float); xf = np.asarray(xf, float)
    B = x0.shape[0]
    fr = (np.arange(n) / (n - 1))[None, :, None]
    x = x0[:, None, :] + fr * (xf - x0)[:, None, :]
    th0 = _wrap(x0[:, 2]); thf = _wrap(xf[:, 2])
    x[:, :, 2] = _wrap(th0[:, None] + fr[:, :, 0] * _wrap(thf - th0)[:, None])
    x[:, 0] = x0; x[:, 0, 2] = th0
    x[:, -1] = xf; x[:, -1, 2] = thf
    return x, np.zeros((B, n, 2)), np.full(B, float(dt_ref))


def travel_direction_guess(x0, xf, n: int, dt_ref: float, reverse: bool = False):
    x0 = np.asarray(x0, float); xf = np.asarray(xf, float)
    B = x0.shape[0]
    fr = (np.arange(n) / (n - 1))[None, :, None]
    x = x0[:, None, :] + fr * (xf - x0)[:, None, :]
    d = xf - x0                                                    # direction over all three components, as the reference normalises it (:158-161)
    nrm = np.linalg.norm(d, axis=1)
    dn = np.where(nrm[:, None] != 0, d / np.where(nrm == 0, 1.0, nrm)[:, None], d)
    orient = np.arctan2(dn[:, 1], dn[:, 0])
    behind = dn[:, 0] * np.cos(x0[:, 2]) + dn[:, 1] * np.sin(x0[:, 2]) < 0
    orient = np.where(behind, _wrap(orient + np.pi), orient)        # :164-172
    if reverse:
        orient = _wrap(orient + np.pi)
    x[:, 1:-1, 2] = orient[:, None]
    x[:, 0] = x0
    x[:, -1] = xf
    return x, np.zeros((B, n, 2)), np.full(B, float(dt_ref))


def select_best(status, dt, n_candidates: int, dt_free: bool = True):
    """status, dt: results of a batch laid out candidate-major ([c * B + b]).  Returns for every instance the index of the winning
    candidate: converged ones first; among them the shortest transition time when dt is a variable (minimum-time objectives), otherwise
    the first converged one; without any converged candidate the first one."""
    status = np.asarray(status).reshape(n_candidates, -1)
    dt = np.asarray(dt, float).reshape(n_candidates, -1)
    ok = status == 0
    key = np.where(ok, dt if dt_free else 0.0, np.inf)
    key = key + np.arange(n_candidates)[:, None] * 1e-12            # ties: the earlier candidate
    best = np.argmin(key, axis=0)
    return np.where(ok.any(axis=0), best, 0)


def solve_best_of(solver, x0, xf, u_prev=None, dt_prev=None, guesses=("cold", "travel")):
    """Solves every instance from each guess in ONE batch (B * len(guesses) instances; the solver must have been created with that
    max_batch) and returns (BatchResult of the winners, winner index per instance, BatchResult of all candidates)."""
    from .solver import BatchResult
    x0 = np.asarray(x0, float); xf = np.asarray(xf, float)
    B, n, dt_ref = x0.shape[0], solver.n, float(solver.cfg.dt_ref)
    gen = {"cold": lambda: cold_start_guess(x0, xf, n, dt_ref), "travel": lambda: travel_direction_guess(x0, xf, n, dt_ref),
           "reverse": lambda: travel_direction_guess(x0, xf, n, dt_ref, reverse=True)}
    parts = [gen[g]() for g in guesses]
    C = len(parts)
    rep = lambda a: None if a is None else np.concatenate([np.asarray(a, float)] * C, 0)
    init = tuple(np.concatenate([p[i] for p in parts], 0) for i in range(3))
    allr = solver.solve(rep(x0), rep(xf), rep(u_prev), rep(dt_prev), init=init)
    win = select_best(allr.status, allr.dt, C, bool(solver.cfg.dt_free))
    idx = win * B + np.arange(B)
    return BatchResult(allr.x[idx], allr.u[idx], allr.dt[idx], allr.status[idx], allr.iters[idx]), win, allr
