"""GPU tests (-m gpu) of the state the handle keeps between control cycles and of the batched closed loop: multipliers carried across
cycles (dual_warm_start), the device-side grid update (warm-start shift on the fixed grid, single-step adaptation + resampling on the
variable grid) and a 50-cycle closed loop of a whole batch that never leaves the device between cycles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("these tests need the MI355X (no HIP device here)")
    torch.zeros(1, device="cuda")
    import mpc_local_planner_amd as pkg
    return pkg, torch


def _advance_carlike(x0, u0, per, L):
    x1 = x0.copy()
    x1[:, 0] += per * u0[:, 0] * np.cos(x0[:, 2]); x1[:, 1] += per * u0[:, 0] * np.sin(x0[:, 2])
    x1[:, 2] = (x0[:, 2] + per * u0[:, 0] * np.tan(u0[:, 1]) / L + np.pi) % (2 * np.pi) - np.pi
    return x1


def test_multipliers_kept_in_the_handle_shorten_the_next_cycle(env, c_oracle):
    """config 2, B = 1024: cycle 1 cold, the plant advances one period, cycle 2 starts from the previous solution.  With dual_warm_start the
    second solve starts from the kept multipliers (mu0 = 1e-3): <= 16 iterations on average (20.5 without), >= 99 % of the previously
    converged instances converge again, same results as the C oracle running the same rule, and mpc_reset forgets the multipliers."""
    from oracle import se2_nlp as R
    from _parity import account
    m, torch = env
    B, n, per = 1024, 50, 0.2
    inputs = m.workloads.carlike_min_time_inputs(B)
    x0, xf, up, dtp = inputs
    ocfg = R.config_carlike_min_time(n)
    oc = c_oracle.from_nlp_config(ocfg)
    s = m.BatchSolver(m.config_carlike_min_time(n, dual_warm_start=True, mu_init_dual=1e-3, mu_init_warm=1e-2), max_batch=B)
    r1 = s.solve(*inputs)
    ds = c_oracle.dual_state(B, n)
    o1 = c_oracle.solve_batch(oc, *inputs, dual_state=ds)
    ok = (r1.status == 0) & (o1[3] == 0) & (np.abs(r1.x - o1[0]).reshape(B, -1).max(1) < 1e-6)          # same first cycle on both sides
    x1 = _advance_carlike(x0, r1.u[:, 0, :], per, 0.4)
    dper = np.full(B, per)
    r2 = s.solve(x1, xf, r1.u[:, 0, :], dper, init=(r1.x, r1.u, r1.dt))
    o2 = c_oracle.solve_batch(oc, x1, xf, r1.u[:, 0, :], dper, init=(r1.x, r1.u, r1.dt), dual_state=ds, dual_mu0=1e-3)
    print(f"[dual warm start] cycle 2: iterations mean {r2.iters[ok].mean():.2f} (oracle {o2[4][ok].mean():.2f}), re-converged {np.mean(r2.status[ok] == 0):.4f}")
    assert r2.iters[ok].mean() <= 17.5 and np.mean(r2.status[ok] == 0) >= 0.99          # r04: 16.6 with the adaptive barrier rule (monotone: 15.5), cold 31.0
    assert abs(r2.iters[ok].mean() - o2[4][ok].mean()) < 0.5
    sub = np.nonzero(ok)[0]
    rr = m.BatchResult(r2.x[sub], r2.u[sub], r2.dt[sub], r2.status[sub], r2.iters[sub])
    account("dual warm start, cycle 2", ocfg, (x1[sub], xf[sub], r1.u[sub, 0, :], dper[sub]), rr, tuple(a[sub] for a in o2))
    # mpc_reset: the same warm solve now starts from re-initialised multipliers (mu_init_warm) and needs more iterations
    s.reset()
    r3 = s.solve(x1, xf, r1.u[:, 0, :], dper, init=(r1.x, r1.u, r1.dt))
    assert r3.iters[ok].mean() > r2.iters[ok].mean() + 2.0
    s.close()


def test_grid_update_on_the_device_is_bit_exact(env):
    """mpc_grid_update_device against the numpy restatements (oracle/se2_nlp.py: warm_start_shifting / find_nearest_state,
    adapt_grid_single_step / resample_trajectory), bit for bit, on solver outputs."""
    from oracle import se2_nlp as R
    m, torch = env
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(961)
    # ---- fixed grid: shift towards the new start
    B, n = 192, 20
    inputs = m.workloads.unicycle_quadratic_inputs(B, seed=962)
    s = m.BatchSolver(m.config_unicycle_quadratic(n), max_batch=B)
    r = s.solve(*inputs)
    adv = rng.integers(0, 4, B)                                     # the robot moved 0..3 grid intervals (+ noise) along its plan
    x0n = r.x[np.arange(B), adv] + rng.normal(0, 0.01, (B, 3))
    dx, du, dd, d0 = T(r.x), T(r.u), T(r.dt), T(x0n)
    s.grid_update_device(B, d0.data_ptr(), dx.data_ptr(), du.data_ptr(), dd.data_ptr())
    s.synchronize()
    gx, gu = dx.cpu().numpy(), du.cpu().numpy()
    shifted = 0
    for b in range(B):
        t = R.warm_start_shifting(R.Trajectory(r.x[b].copy(), r.u[b, :-1].copy(), float(r.dt[b])), x0n[b])
        np.testing.assert_array_equal(gx[b], t.x)
        np.testing.assert_array_equal(gu[b, :-1], t.u)
        np.testing.assert_array_equal(gu[b, -1], t.u[-1])
        shifted += int(not np.array_equal(t.x, r.x[b]))
    assert shifted > B // 2
    s.close()
    # ---- variable grid: n + 1 / n - 1 / unchanged, resampled
    B, n = 192, 40
    inputs = m.workloads.carlike_min_time_inputs(B, seed=963, goal_range=(1.0, 6.0))
    ocfg = R.config_carlike_min_time(n)
    s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    n0 = rng.integers(20, 36, B).astype(np.int32)
    s.set_grid_sizes(n0)
    r = s.solve(*inputs)
    dx, du, dd = T(r.x), T(r.u), T(r.dt)
    s.grid_update_device(B, None, dx.data_ptr(), du.data_ptr(), dd.data_ptr(), adapt=True, n_min=3, n_max=n, dt_hyst_ratio=0.1)
    s.synchronize()
    gx, gu, gd, gn = dx.cpu().numpy(), du.cpu().numpy(), dd.cpu().numpy(), s.grid_sizes(B)
    changed = 0
    for b in range(B):
        k = int(n0[b])
        t = R.adapt_grid_single_step(ocfg, R.Trajectory(r.x[b, :k].copy(), r.u[b, :k - 1].copy(), float(r.dt[b])), n_min=3, n_max=n, hyst=0.1)
        kn = t.x.shape[0]
        assert gn[b] == kn
        np.testing.assert_array_equal(gx[b, :kn], t.x)
        np.testing.assert_array_equal(gu[b, :kn - 1], t.u)
        assert gd[b] == t.dt
        changed += int(kn != k)
    assert changed > B // 4 and (gn > n0).any() and (gn < n0).any()
    s.close()


@pytest.mark.parametrize("dual", [False, True])
def test_batched_closed_loop_50_cycles_on_the_device(env, c_oracle, dual):
    """SURVEY 8c level 3 for a batch: 128 unicycle planners (config 1 family, fixed grid, moving-horizon shift) run 50 control cycles; between
    cycles only device arrays are touched (plant step with torch on the GPU, mpc_grid_update_device, the next solve reads the previous
    outputs).  Every cycle's inputs are also given to the C oracle (same shift restated in numpy): the device must reproduce the oracle's
    commands cycle by cycle, re-converge throughout and drive the robots to their goals.  dual = False: both sides run the identical algorithm
    (commands equal to round-off); dual = True: the device additionally starts every cycle from the multipliers it kept (shifted with the
    trajectory) -- another iterate path to the same KKT points, fewer iterations."""
    from oracle import se2_nlp as R
    m, torch = env
    dev = torch.device("cuda", 0)
    B, n, per, cycles = 128, 20, 0.3, 50
    x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(B, seed=971, goal_range=(1.0, 2.0))
    ocfg = R.config_unicycle_quadratic(n)
    oc = c_oracle.from_nlp_config(ocfg)
    s = m.BatchSolver(m.config_unicycle_quadratic(n, dual_warm_start=dual), max_batch=B)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dx0, dxf, dup, ddtp = T(x0), T(xf), T(up), T(np.full(B, per))
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    xi = torch.empty_like(xo); ui = torch.empty_like(uo); di = torch.empty_like(do)
    worst, conv_min, iters = 0.0, 1.0, []
    for c in range(cycles):
        init = None if c == 0 else (xi.data_ptr(), ui.data_ptr(), di.data_ptr())
        s.solve_device(B, dx0.data_ptr(), dxf.data_ptr(), dup.data_ptr(), ddtp.data_ptr(), *(init or (None, None, None)), xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize()
        # checker (host copies are for the ORACLE only; the loop itself continues from the device arrays)
        hx0, hup = dx0.cpu().numpy(), dup.cpu().numpy()
        hinit = None if c == 0 else (xi.cpu().numpy(), ui.cpu().numpy(), di.cpu().numpy())
        o = c_oracle.solve_batch(oc, hx0, xf, hup, np.full(B, per), init=hinit)
        hu, hs = uo.cpu().numpy(), st.cpu().numpy()
        both = (hs == 0) & (o[3] == 0)
        conv_min = min(conv_min, float((hs == 0).mean()))
        worst = max(worst, float(np.abs(hu[both, 0] - o[1][both, 0]).max()))
        iters.append(float(it.float().mean().item()))
        assert (hs == o[3]).mean() > 0.98
        # plant: unicycle, one period with the first command (on the device)
        u0 = uo[:, 0, :]
        dx0 = torch.stack([dx0[:, 0] + per * u0[:, 0] * torch.cos(dx0[:, 2]), dx0[:, 1] + per * u0[:, 0] * torch.sin(dx0[:, 2]),
                           torch.remainder(dx0[:, 2] + per * u0[:, 1] + np.pi, 2 * np.pi) - np.pi], 1).contiguous()
        dup = u0.clone()
        xi.copy_(xo); ui.copy_(uo); di.copy_(do)
        s.grid_update_device(B, dx0.data_ptr(), xi.data_ptr(), ui.data_ptr(), di.data_ptr())
    print(f"[closed loop] {cycles} cycles x {B} planners: worst |u0(device) - u0(oracle)| {worst:.2e}, lowest converged fraction {conv_min:.3f}, "
          f"iterations first / later cycles {iters[0]:.1f} / {np.mean(iters[5:]):.1f}")
    assert worst < (1e-3 if dual else 1e-6) and conv_min > 0.97
    assert np.mean(iters[5:]) < (9.0 if dual else 14.0)
    dist = torch.linalg.norm(dx0[:, :2] - dxf[:, :2], dim=1).cpu().numpy()
    assert np.median(dist) < 0.15 and (dist < 0.4).mean() > 0.9
    assert np.mean(iters[5:]) < iters[0]
    s.close()


def test_mixed_precision_meets_the_fp64_tolerance_on_config5(env, c_oracle):
    """BASELINE.json configs[4] shape (kinematic bicycle, variable-dt time-optimal, n = 120): MPC_MIXED = fp32 main phase (to 1e-4) + fp64
    refinement started from its iterate and multipliers.  Against the C oracle (fp64): median trajectory difference < 1e-4 (it is ~1e-8),
    converged fraction not below the fp64 kernel's by more than 2 points, and EVERY converged result is accounted for: within 1e-4 of the oracle
    or -- where the fp32 phase wandered into another basin of this multi-modal NLP (plain fp32 does the same) -- a KKT point of the
    reference-form NLP on its own (feasibility / stationarity / complementarity <= 1e-6 in fp64).  Plain fp32 (tol 1e-4) is shown beside it."""
    from oracle import se2_nlp as R
    from mpc_local_planner_amd import _abi as A
    from _parity import account
    m, torch = env
    B, n = 256, 120
    inputs = m.workloads.bicycle_min_time_inputs(B)
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(R.config_bicycle_min_time(n)), *inputs)
    out = {}
    for tag, kw in (("fp64", dict(precision=A.FP64)), ("mixed", dict(precision=A.MIXED)), ("fp32", dict(precision=A.FP32, tol=1e-4))):
        s = m.BatchSolver(m.config_bicycle_min_time(n, **kw), max_batch=B)
        r = s.solve(*inputs)
        r = s.solve(*inputs)
        both = (r.status == 0) & (ref[3] == 0)
        err = np.abs(r.x - ref[0]).reshape(B, -1).max(1)[both]
        out[tag] = (float((r.status == 0).mean()), float(np.median(err)), float(np.percentile(err, 95)), s.last_kernel_ms(), float(r.iters.mean()), float((err < 1e-3).mean()))
        print(f"[config 5, {tag}] converged {out[tag][0]:.3f} (oracle {np.mean(ref[3] == 0):.3f}); |x - oracle| median {out[tag][1]:.1e}, p95 {out[tag][2]:.1e}, "
              f"within 1e-3: {out[tag][5]:.3f}; kernel {out[tag][3]:.2f} ms; iterations {out[tag][4]:.1f}")
        if tag == "mixed":
            account("config 5 shape, mixed precision", R.config_bicycle_min_time(n), inputs, r, ref)
        s.close()
    assert out["mixed"][1] < 1e-4 and out["mixed"][5] > 0.85
    assert out["mixed"][0] >= out["fp64"][0] - 0.02
    assert out["fp64"][1] < 1e-6


def test_config5_candidates_vs_oracle_rule_fp64_and_mixed(env, c_oracle):
    """BASELINE.json configs[4] shape (kinematic bicycle, n = 120) with the candidate set of its bench leg (reference cold start, travel, travel-reverse,
    Hermite FF; caps 60/50/45/40): the identical rule on the C oracle against the device in fp64 (winners and trajectories) and in MPC_MIXED (fp32
    main phase picks the winner, fp64 refinement), accounting for every converged device result: within 1e-4 of the oracle rule's result or a KKT point
    of the reference-form NLP on its own; nothing unclassified."""
    from oracle import se2_nlp as R, candidates as OC
    from mpc_local_planner_amd import _abi as A
    from _parity import account
    m, torch = env
    B, n = 1024, 120          # the batch of the bench leg (per-GPU share of BASELINE configs[4])
    kinds, caps, pars = (A.CAND_REFERENCE, A.CAND_TRAVEL, A.CAND_TRAVEL_REVERSE, A.CAND_HERMITE_FF), (60, 50, 45, 40), (0.0, 0.0, 0.0, 2.0)
    ocfg = R.config_bicycle_min_time(n)
    inputs = m.workloads.bicycle_min_time_inputs(B)
    ox, ou, od, ost, oit, owin, olow, allr = OC.solve_candidates(c_oracle, lambda cap: c_oracle.from_nlp_config(ocfg, max_iter=cap), *inputs, kinds, caps, n, ocfg.dt_ref, params=pars)
    for tag, prec in (("fp64", A.FP64), ("mixed", A.MIXED)):
        s = m.BatchSolver(m.config_bicycle_min_time(n, precision=prec, candidates=kinds, candidate_max_iter=caps, candidate_param=pars), max_batch=B)
        r = s.solve(*inputs)
        win, tot = s.last_candidates(B)
        conv = r.status == 0
        print(f"[config 5 candidates, {tag}] device converged {conv.mean():.4f} (oracle rule {np.mean(ost == 0):.4f}); winners device {np.bincount(win + 1, minlength=5).tolist()} "
              f"oracle {np.bincount(owin + 1, minlength=5).tolist()} (index 0 = none); equal winners {np.mean(win == owin):.4f}")
        assert conv.mean() >= 0.97
        if tag == "fp64":
            assert np.mean(win == owin) > 0.95
        match, other = account(f"config 5 shape with candidates, {tag}", ocfg, inputs, r, (ox, ou, od, ost, oit))
        # fp64 is the precision of the bench leg that counts for BASELINE configs[4] (r05: affordable since the factorisation data left LDS): >= 95 % of the 1024
        # answers within 1e-4 of the oracle rule's (measured 1022).  MPC_MIXED keeps its 60 % floor (833: the fp32 phase picks another basin in ~19 %).
        assert match.sum() > (0.95 if tag == "fp64" else 0.6) * B
        s.close()


def test_step_batch_is_the_separate_calls_in_one(env):
    """mpc_step_batch = PredictiveController::step's outer OCP iterations (src/controller.cpp:70-72,172) enqueued at once: three x (grid update -> solve)
    of 96 car-like planners on the variable grid with adaptation give, bit for bit, what solve / grid update / solve / grid update / solve give through the
    separate entry points -- trajectories, dt, status, iteration counts of the last solve and the adapted grid sizes."""
    m, torch = env
    dev = torch.device("cuda", 0)
    B, n = 96, 30
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=981, goal_range=(1.0, 4.0))
    mk = lambda: m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    # one call
    s1 = mk()
    r1, ng1 = s1.step(x0, xf, up, dtp, outer_iterations=3, adapt=True, n_min=3, n_max=n, dt_hyst_ratio=0.1)
    s1.close()
    # separate calls (device entry points)
    s2 = mk()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = [T(x0), T(xf), T(up), T(dtp)]
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    args = (B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr())
    outs = (xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
    s2.solve_device(*args, None, None, None, *outs)
    for _ in range(2):
        s2.grid_update_device(B, d[0].data_ptr(), xo.data_ptr(), uo.data_ptr(), do.data_ptr(), adapt=True, n_min=3, n_max=n, dt_hyst_ratio=0.1)
        s2.solve_device(*args, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), *outs)
    s2.synchronize()
    ng2 = s2.grid_sizes(B)
    s2.close()
    np.testing.assert_array_equal(ng1, ng2)
    np.testing.assert_array_equal(r1.status, st.cpu().numpy()); np.testing.assert_array_equal(r1.iters, it.cpu().numpy())
    np.testing.assert_array_equal(r1.dt, do.cpu().numpy())
    for b in range(B):
        k = int(ng1[b])
        np.testing.assert_array_equal(r1.x[b, :k], xo[b, :k].cpu().numpy()); np.testing.assert_array_equal(r1.u[b, :k], uo[b, :k].cpu().numpy())
    assert (ng1 != n).sum() > B // 4 and (r1.status == 0).mean() > 0.9          # the adaptation did something, the last solves converged
