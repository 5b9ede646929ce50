"""ORACLE (test infrastructure only): candidate initial trajectories and the winner rule, restated on the CPU.

The product generates the candidates on the device (mpc_wave.hpp::seed_start) and applies the rule with atomics inside the solve
kernel (mpc_capi.hip, "exit protocol"); this file restates both in numpy so that tests can run the identical rule on the C oracle:

  kinds (include/mpc_hip.h, enum mpc_candidate_kind)
    0 REFERENCE        the 2-pose-plan cold start of Controller::step (src/controller.cpp:807-857 +
                       src/optimal_control/full_discretization_grid_base_se2.cpp:192-239): straight line, shortest-arc heading
    1 TRAVEL           initializeSequences without xinit (...grid_base_se2.cpp:136-190): heading = direction of travel, turned by pi
                       when the goal lies behind the start pose (:164-172)
    2 TRAVEL_REVERSE   the other driving direction (ours)
    3 BLEND            TRAVEL with the heading turned from the start heading over the first m grid points and into the goal heading
                       over the last m (ours)
    4 BLEND_REVERSE    the same around TRAVEL_REVERSE (ours)
  rule: the candidate with the LOWEST index that converges within its own iteration cap supplies the result; without any, candidate 0's
        last iterate and status are returned.
"""
from __future__ import annotations

import numpy as np

REFERENCE, TRAVEL, TRAVEL_REVERSE, BLEND, BLEND_REVERSE = range(5)


def wrap(th):
    """normalize_theta (include/mpc_local_planner/utils/math_utils.h:81-91), vectorised"""
    th = np.asarray(th, float)
    out = th - np.floor(th / (2 * np.pi)) * 2 * np.pi
    out = np.where(out >= np.pi, out - 2 * np.pi, out)
    out = np.where(out < -np.pi, out + 2 * np.pi, out)
    return np.where((th >= -np.pi) & (th < np.pi), th, out)


def guess(kind: int, x0, xf, n: int, dt_ref: float, blend: int = 8):
    """(x (B,n,3), u (B,n,2) = 0, dt (B,)) of candidate `kind`; controls zero = the solver seeds them from the states."""
    x0 = np.asarray(x0, float).copy(); xf = np.asarray(xf, float).copy()
    x0[:, 2] = wrap(x0[:, 2]); xf[:, 2] = wrap(xf[:, 2])
    B = x0.shape[0]
    fr = (np.arange(n) / (n - 1))[None, :]
    x = np.empty((B, n, 3))
    x[:, :, 0] = x0[:, None, 0] + fr * (xf[:, 0] - x0[:, 0])[:, None]
    x[:, :, 1] = x0[:, None, 1] + fr * (xf[:, 1] - x0[:, 1])[:, None]
    if kind == REFERENCE:
        x[:, :, 2] = wrap(x0[:, None, 2] + fr * wrap(xf[:, 2] - x0[:, 2])[:, None])
    else:
        dx, dy = xf[:, 0] - x0[:, 0], xf[:, 1] - x0[:, 1]
        orient = np.arctan2(dy, dx)
        behind = dx * np.cos(x0[:, 2]) + dy * np.sin(x0[:, 2]) < 0                 # :164-172
        orient = np.where(behind, wrap(orient + np.pi), orient)
        if kind in (TRAVEL_REVERSE, BLEND_REVERSE):
            orient = wrap(orient + np.pi)
        x[:, :, 2] = orient[:, None]
        if kind in (BLEND, BLEND_REVERSE):
            m = min(int(blend), (n - 1) // 2)
            d0, df = wrap(orient - x0[:, 2]), wrap(orient - xf[:, 2])
            for k in range(1, n - 1):
                if k < m:
                    x[:, k, 2] = wrap(x0[:, 2] + (k / m) * d0)
                if n - 1 - k < m:
                    x[:, k, 2] = wrap(xf[:, 2] + ((n - 1 - k) / m) * df)
    x[:, 0] = x0
    x[:, -1] = xf
    return x, np.zeros((B, n, 2)), np.full(B, float(dt_ref))


def apply_rule(status, n_candidates: int):
    """status laid out [c][b] -> winner index per instance (-1: none converged)."""
    ok = np.asarray(status).reshape(n_candidates, -1) == 0
    return np.where(ok.any(axis=0), np.argmax(ok, axis=0), -1)


def solve_candidates(c_oracle, ocfg_for_cap, x0, xf, u_prev, dt_prev, kinds, caps, n: int, dt_ref: float, blend: int = 8, init=None):
    """Runs every candidate of every instance on the C oracle (oracle/mpc_oracle.c) to its own iteration cap and applies the rule.
    ocfg_for_cap(cap) -> OracleConfig with max_iter = cap.  `init` (x, u, dt) replaces the REFERENCE candidate's guess (warm start).
    Returns (x, u, dt, status, iters of the winner / of candidate 0 when none converged, winner, iters_total, all per-candidate results)."""
    res = []
    for kind, cap in zip(kinds, caps):
        oc = ocfg_for_cap(cap)
        if kind == REFERENCE:
            r = c_oracle.solve_batch(oc, x0, xf, u_prev, dt_prev, init=init)      # the oracle's own cold start / the given guess
        else:
            r = c_oracle.solve_batch(oc, x0, xf, u_prev, dt_prev, init=guess(kind, x0, xf, n, dt_ref, blend))
        res.append(r)
    st = np.stack([r[3] for r in res])
    win = apply_rule(st, len(kinds))
    src = np.where(win >= 0, win, 0)
    b = np.arange(st.shape[1])
    pick = lambda j: np.stack([r[j] for r in res])[src, b]
    # a hedge is stopped as soon as a higher-priority candidate converges, so total iterations depend on timing on the device; the
    # timing-independent part is: every candidate ahead of the winner ran to its cap or to its own failure, the winner ran to convergence
    it = np.stack([r[4] for r in res])
    lower = np.where(np.arange(len(kinds))[:, None] <= src[None, :], it, 0).sum(0)
    return pick(0), pick(1), pick(2), pick(3), pick(4), win, lower, res
