"""Host-side candidate initial trajectories (mpc_local_planner_amd/candidates.py): the guesses against the oracle's restatements of the
reference's two initialisations, the winner selection, and -- with the C oracle standing in for the device behind the same solve()
signature -- the effect on the config-2 workload.  CPU only."""
import numpy as np
import pytest

from oracle import se2_nlp as R
import mpc_local_planner_amd as m
from mpc_local_planner_amd import candidates as K


def test_guesses_match_the_reference_initialisations():
    x0, xf, _, _ = m.workloads.carlike_min_time_inputs(32, seed=3)
    cfg = R.config_carlike_min_time(30)
    xc, uc, dc = K.cold_start_guess(x0, xf, 30, cfg.dt_ref)
    xt, ut, dt = K.travel_direction_guess(x0, xf, 30, cfg.dt_ref)
    xr, _, _ = K.travel_direction_guess(x0, xf, 30, cfg.dt_ref, reverse=True)
    for b in range(32):
        np.testing.assert_allclose(xc[b], R.cold_start(cfg, x0[b], xf[b]).x, atol=1e-12)                      # a2 + a5 (2-pose plan)
        ref = R.initialize_sequences_straight_line(cfg, x0[b], xf[b]).x                                        # a4
        np.testing.assert_allclose(xt[b], ref, atol=1e-12)
        np.testing.assert_allclose(np.abs(R.normalize_theta(xr[b, 1:-1, 2] - ref[1:-1, 2])), np.pi, atol=1e-12)
        np.testing.assert_array_equal(xr[b, [0, -1]], ref[[0, -1]])
    assert (uc == 0).all() and (ut == 0).all() and (dc == cfg.dt_ref).all() and (dt == cfg.dt_ref).all()


def test_select_best_prefers_converged_then_shortest_time():
    #            inst 0      1      2      3
    status = [0, 1, 0, 1,   0, 0, 1, 1]            # candidate 0, candidate 1
    dt = [0.5, 0.2, 0.3, 0.1,   0.4, 0.9, 0.2, 0.05]
    np.testing.assert_array_equal(K.select_best(status, dt, 2, dt_free=True), [1, 1, 0, 0])
    np.testing.assert_array_equal(K.select_best(status, dt, 2, dt_free=False), [0, 1, 0, 0])


class _OracleBackedSolver:
    """stands in for BatchSolver in this CPU test: same solve() signature and result type, the C oracle does the work"""
    def __init__(self, ocfg, c_oracle):
        self.ocfg, self.co, self.n = ocfg, c_oracle, ocfg.n
        self.cfg = type("Cfg", (), {"dt_ref": ocfg.dt_ref, "dt_free": ocfg.dt_free})()

    def solve(self, x0, xf, u_prev=None, dt_prev=None, init=None, obstacles=None):
        xo, uo, do, st, it = self.co.solve_batch(self.co.from_nlp_config(self.ocfg), x0, xf, u_prev, dt_prev, init=init)
        return m.BatchResult(xo, uo, do, st, it)


def test_best_of_two_guesses_on_config2(c_oracle):
    """config-2 workload (car-like minimum time, n = 50): the reference's two initialisations as candidates of every instance.  The
    converged fraction rises from ~0.94 to > 0.99 and no instance ends with a longer transition time than from the cold start alone."""
    B = 192
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = _OracleBackedSolver(R.config_carlike_min_time(50), c_oracle)
    best, win, allr = K.solve_best_of(s, x0, xf, up, dtp, guesses=("cold", "travel"))
    single = s.solve(x0, xf, up, dtp)                                   # device-style cold start (no init)
    same = (allr.status[:B] == 0) & (single.status == 0)
    err = np.abs(allr.x[:B, :, :2] - single.x[:, :, :2]).reshape(B, -1).max(1)[same]
    assert (allr.status[:B] == single.status).mean() > 0.97 and np.median(err) < 1e-7 and (err < 1e-3).mean() > 0.95   # candidate 0 IS the cold start
    assert (best.status == 0).mean() > 0.99 > (allr.status[:B] == 0).mean()
    ok = allr.status[:B] == 0                                          # never worse than the cold-start candidate of the same batch
    assert (best.status[ok] == 0).all() and (best.dt[ok] <= allr.dt[:B][ok] + 1e-12).all()
    assert (best.dt[ok] < allr.dt[:B][ok] - 1e-4).mean() > 0.1          # ... and strictly better in a good share of the instances
    assert (win == 1).mean() > 0.1                                      # the second guess does win a share of the instances


def test_oracle_candidate_kinds_restate_the_guesses():
    """oracle/candidates.py (what the GPU tests check the device-side candidates against) vs the reference initialisations restated in
    oracle/se2_nlp.py and the host-side guesses of the product package."""
    from oracle import candidates as OC
    x0, xf, _, _ = m.workloads.carlike_min_time_inputs(48, seed=5)
    n, cfg = 30, R.config_carlike_min_time(30)
    for b in range(48):
        np.testing.assert_allclose(OC.guess(OC.REFERENCE, x0[b:b + 1], xf[b:b + 1], n, cfg.dt_ref)[0][0], R.cold_start(cfg, x0[b], xf[b]).x, atol=1e-12)
        np.testing.assert_allclose(OC.guess(OC.TRAVEL, x0[b:b + 1], xf[b:b + 1], n, cfg.dt_ref)[0][0], R.initialize_sequences_straight_line(cfg, x0[b], xf[b]).x, atol=1e-12)
    np.testing.assert_allclose(OC.guess(OC.TRAVEL_REVERSE, x0, xf, n, 0.3)[0], K.travel_direction_guess(x0, xf, n, 0.3, reverse=True)[0], atol=1e-12)
    # blend: start / goal poses exact, the middle of the horizon carries the travel direction, the ends turn monotonically from / into the pose headings
    for kind, base in ((OC.BLEND, OC.TRAVEL), (OC.BLEND_REVERSE, OC.TRAVEL_REVERSE)):
        xb = OC.guess(kind, x0, xf, n, 0.3, blend=8)[0]
        xt = OC.guess(base, x0, xf, n, 0.3)[0]
        np.testing.assert_array_equal(xb[:, :, :2], xt[:, :, :2])
        np.testing.assert_array_equal(xb[:, 8:n - 8, 2], xt[:, 8:n - 8, 2])
        d0 = OC.wrap(xt[:, 10, 2] - x0[:, 2])
        for k in range(1, 8):
            np.testing.assert_allclose(OC.wrap(xb[:, k, 2] - x0[:, 2]), k / 8 * d0, atol=1e-12)
            np.testing.assert_allclose(OC.wrap(xb[:, n - 1 - k, 2] - xf[:, 2]), k / 8 * OC.wrap(xt[:, 10, 2] - xf[:, 2]), atol=1e-12)


def test_candidate_rule_lowest_converged_index_wins():
    from oracle import candidates as OC
    #          inst 0  1  2  3
    status = [[0, 1, 1, 1], [0, 0, 1, 2], [1, 0, 0, 3]]
    np.testing.assert_array_equal(OC.apply_rule(status, 3), [0, 1, 2, -1])


def test_candidate_rule_on_config2_with_the_c_oracle(c_oracle):
    """the rule the device applies (reference cold start first, two blended-heading hedges, every candidate capped at 60 iterations) on the
    config-2 workload: > 98 % of the instances end converged and every instance the capped reference path solves keeps that answer."""
    from oracle import candidates as OC
    B, n = 256, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    ocfg = R.config_carlike_min_time(n)
    kinds, caps = (OC.REFERENCE, OC.BLEND, OC.BLEND_REVERSE), (60, 60, 60)
    x, u, dt, st, it, win, low, allr = OC.solve_candidates(c_oracle, lambda cap: c_oracle.from_nlp_config(ocfg, max_iter=cap), x0, xf, up, dtp, kinds, caps, n, ocfg.dt_ref)
    assert (st == 0).mean() > 0.98 > (allr[0][3] == 0).mean()          # r04: 98.4 % with these two blended hedges (the headline's Hermite set: 99.9 %)
    ref_ok = allr[0][3] == 0
    assert (win[ref_ok] == 0).all() and np.array_equal(x[ref_ok], allr[0][0][ref_ok])
    assert (it[st == 0] <= 60).all() and (win >= 1).mean() > 0.04          # r04 with the inertia test: the reference path alone solves 92 % within 60 iterations (84 % before), 8 % are answered by a hedge


def test_hermite_candidates_are_smooth_curves_between_the_poses():
    """kinds 5..8 (ours): cubic Hermite curve from the start pose to the goal pose, heading along the tangent (+ pi where the robot drives
    backwards).  End poses exact, the curve leaves / arrives along the pose headings, forward and reverse variants mirror each other's heading."""
    from oracle import candidates as OC
    x0, xf, _, _ = m.workloads.carlike_min_time_inputs(64, seed=6)
    n = 50
    ff = OC.guess(OC.HERMITE_FF, x0, xf, n, 0.3, param=2.0)[0]
    rr = OC.guess(OC.HERMITE_RR, x0, xf, n, 0.3, param=2.0)[0]
    fr = OC.guess(OC.HERMITE_FR, x0, xf, n, 0.3, param=2.0)[0]
    for g in (ff, rr, fr):
        np.testing.assert_array_equal(g[:, 0], np.c_[x0[:, :2], OC.wrap(x0[:, 2])])
        np.testing.assert_array_equal(g[:, -1], np.c_[xf[:, :2], OC.wrap(xf[:, 2])])
    # forward-forward: the first step leaves along the start heading, the last one arrives along the goal heading
    d0 = ff[:, 1, :2] - ff[:, 0, :2]; d1 = ff[:, -1, :2] - ff[:, -2, :2]
    assert (np.abs(OC.wrap(np.arctan2(d0[:, 1], d0[:, 0]) - x0[:, 2])) < 0.15).all() and (np.abs(OC.wrap(np.arctan2(d1[:, 1], d1[:, 0]) - xf[:, 2])) < 0.15).all()
    # the heading of the guess is the tangent direction (forward) / its opposite (reverse)
    t = ff[:, 11, :2] - ff[:, 9, :2]
    assert (np.abs(OC.wrap(ff[:, 10, 2] - np.arctan2(t[:, 1], t[:, 0]))) < 0.05).all()
    t = rr[:, 11, :2] - rr[:, 9, :2]
    assert (np.abs(OC.wrap(rr[:, 10, 2] - np.arctan2(t[:, 1], t[:, 0]) - np.pi)) < 0.05).all()
    # forward-reverse: forward in the first half, backwards in the second
    t = fr[:, 41, :2] - fr[:, 39, :2]
    assert (np.abs(OC.wrap(fr[:, 40, 2] - np.arctan2(t[:, 1], t[:, 0]) - np.pi)) < 0.05).all()
    np.testing.assert_array_equal(fr[:, :24, :2], OC.guess(OC.HERMITE_FR, x0, xf, n, 0.3, param=2.0)[0][:, :24, :2])


def test_hermite_hedges_on_config2_with_the_c_oracle(c_oracle):
    """bench.py's candidate set (reference cold start + three Hermite hedges, caps 60 / 45 / 40 / 35) on the config-2 workload with the C oracle:
    >= 99 % converged, the reference path keeps every instance it solves within its cap."""
    from oracle import candidates as OC
    B, n = 256, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    ocfg = R.config_carlike_min_time(n)
    kinds, caps, pars = (OC.REFERENCE, OC.HERMITE_FF, OC.HERMITE_FF, OC.HERMITE_FR), (60, 45, 40, 35), (0.0, 2.0, 3.0, 1.5)
    x, u, dt, st, it, win, low, allr = OC.solve_candidates(c_oracle, lambda cap: c_oracle.from_nlp_config(ocfg, max_iter=cap), x0, xf, up, dtp, kinds, caps, n, ocfg.dt_ref, params=pars)
    assert (st == 0).mean() >= 0.99
    ref_ok = allr[0][3] == 0
    assert (win[ref_ok] == 0).all() and np.array_equal(x[ref_ok], allr[0][0][ref_ok])
