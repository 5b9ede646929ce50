"""ORACLE (test infrastructure only; parity unpinned: the reference has no recorded outputs for this path): candidate initial trajectories and the winner rule, restated on the CPU.

The product generates the candidates on the device (mpc_wave.hpp::seed_start) and applies the rule with atomics inside the solve
kernel (mpc_solve_kernel.hpp, "exit protocol"); this file restates both in numpy so that tests can run the identical rule on the C oracle:

  kinds (include/mpc_hip.h, enum mpc_candidate_kind)
    0 REFERENCE        the 2-pose-plan cold start of Controller::step (src/controller.cpp:807-857 +
                       src/optimal_control/full_discretization_grid_base_se2.cpp:192-239): straight line, shortest-arc heading
    1 TRAVEL           initializeSequences without xinit (...grid_base_se2.cpp:136-190): heading = direction of travel, turned by pi
                       when the goal lies behind the start pose (:164-172)
    2 TRAVEL_REVERSE   the other driving direction (ours)
    3 BLEND            TRAVEL with the heading turned from the start heading over the first m grid points and into the goal heading
                       over the last m (ours)
    4 BLEND_REVERSE    the same around TRAVEL_REVERSE (ours)
    5..8 HERMITE_FF / _RR / _FR / _RF   (ours) positions on the cubic Hermite curve from the start pose to the goal pose whose end tangents
                       point along the two headings, scaled by param * |goal - start| and signed by the driving direction at that end
                       (F forward, R reverse: first letter = start, second = goal); heading = tangent direction, + pi where the robot drives
                       backwards (the first half of the horizon takes the start's direction, the second half the goal's)
  rule: the candidate with the LOWEST index that converges within its own iteration cap supplies the result; without any, candidate 0's
        last iterate and status are returned.
"""
from __future__ import annotations

import numpy as np

REFERENCE, TRAVEL, TRAVEL_REVERSE, BLEND, BLEND_REVERSE, HERMITE_FF, HERMITE_RR, HERMITE_FR, HERMITE_RF = range(9)
HERMITE_SIGNS = {HERMITE_FF: (1.0, 1.0), HERMITE_RR: (-1.0, -1.0), HERMITE_FR: (1.0, -1.0), HERMITE_RF: (-1.0, 1.0)}


def wrap(th):
    """normalize_theta (include/mpc_local_planner/utils/math_utils.h:81-91), vectorised"""
    th = np.asarray(th, float)
    out = th - np.floor(th / (2 * np.pi)) * 2 * np.pi
    out = np.where(out >= np.pi, out - 2 * np.pi, out)
    out = np.where(out < -np.pi, out + 2 * np.pi, out)
    return np.where((th >= -np.pi) & (th < np.pi), th, out)


def guess(kind: int, x0, xf, n: int, dt_ref: float, blend: int = 8, param: float = 0.0):
    """(x (B,n,3), u (B,n,2) = 0, dt (B,)) of candidate `kind`; controls zero = the solver seeds them from the states.
    param: tangent scale of the HERMITE kinds (0 -> 2.0)."""
    x0 = np.asarray(x0, float).copy(); xf = np.asarray(xf, float).copy()
    x0[:, 2] = wrap(x0[:, 2]); xf[:, 2] = wrap(xf[:, 2])
    B = x0.shape[0]
    fr = (np.arange(n) / (n - 1))[None, :]
    if kind in HERMITE_SIGNS:
        s0, s1 = HERMITE_SIGNS[kind]
        c = param if param > 0 else 2.0
        t = fr
        dx, dy = xf[:, 0] - x0[:, 0], xf[:, 1] - x0[:, 1]
        d = np.sqrt(dx * dx + dy * dy)
        m0x, m0y = s0 * c * d * np.cos(x0[:, 2]), s0 * c * d * np.sin(x0[:, 2])
        m1x, m1y = s1 * c * d * np.cos(xf[:, 2]), s1 * c * d * np.sin(xf[:, 2])
        t2, t3 = t * t, t * t * t
        h00, h10, h01, h11 = 2 * t3 - 3 * t2 + 1, t3 - 2 * t2 + t, -2 * t3 + 3 * t2, t3 - t2
        g00, g10, g01, g11 = 6 * t2 - 6 * t, 3 * t2 - 4 * t + 1, -6 * t2 + 6 * t, 3 * t2 - 2 * t
        x = np.empty((B, n, 3))
        x[:, :, 0] = h00 * x0[:, None, 0] + h10 * m0x[:, None] + h01 * xf[:, None, 0] + h11 * m1x[:, None]
        x[:, :, 1] = h00 * x0[:, None, 1] + h10 * m0y[:, None] + h01 * xf[:, None, 1] + h11 * m1y[:, None]
        tx = g00 * x0[:, None, 0] + g10 * m0x[:, None] + g01 * xf[:, None, 0] + g11 * m1x[:, None]
        ty = g00 * x0[:, None, 1] + g10 * m0y[:, None] + g01 * xf[:, None, 1] + g11 * m1y[:, None]
        th = np.arctan2(ty, tx)
        back = np.where(2 * np.arange(n)[None, :] < n - 1, s0, s1) < 0
        x[:, :, 2] = np.where(back, wrap(th + np.pi), th)
        x[:, 0] = x0
        x[:, -1] = xf
        return x, np.zeros((B, n, 2)), np.full(B, float(dt_ref))
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


def solve_candidates(c_oracle, ocfg_for_cap, x0, xf, u_prev, dt_prev, kinds, caps, n: int, dt_ref: float, blend: int = 8, init=None, params=None):
    """Runs every candidate of every instance on the C oracle (oracle/mpc_oracle.c) to its own iteration cap and applies the rule.
    ocfg_for_cap(cap) -> OracleConfig with max_iter = cap.  `init` (x, u, dt) replaces the REFERENCE candidate's guess (warm start).
    Returns (x, u, dt, status, iters of the winner / of candidate 0 when none converged, winner, iters_total, all per-candidate results)."""
    res = []
    params = list(params) if params is not None else [0.0] * len(kinds)
    for kind, cap, par in zip(kinds, caps, params):
        oc = ocfg_for_cap(cap)
        if kind == REFERENCE:
            r = c_oracle.solve_batch(oc, x0, xf, u_prev, dt_prev, init=init)      # the oracle's own cold start / the given guess
        else:
            r = c_oracle.solve_batch(oc, x0, xf, u_prev, dt_prev, init=guess(kind, x0, xf, n, dt_ref, blend, par))
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
