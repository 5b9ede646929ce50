"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle
(C restatement with banded LU; numpy dense IPM for the golden fixtures), plus size-independent
properties at BASELINE.json's full batch size.  Tolerances: fp64 trajectories within 1e-6 of the
oracle where both follow the same iterate sequence (north-star bound: 1e-4); fp32 within 1e-2."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("these tests need the MI355X (no HIP device here)")
    torch.zeros(1, device="cuda")          # torch's HIP runtime before the library's
    import mpc_local_planner_amd as pkg
    return pkg


def _cases(m):
    from oracle import se2_nlp as R
    return {
        "carlike_min_time_n50": (m.config_carlike_min_time(50), R.config_carlike_min_time(50)),
        "carlike_min_time_n20": (m.config_carlike_min_time(20), R.config_carlike_min_time(20)),
        "unicycle_quadratic_n20": (m.config_unicycle_quadratic(20), R.config_unicycle_quadratic(20)),
        "bicycle_min_time_n30": (m.config_bicycle_min_time(30), R.config_bicycle_min_time(30)),
    }


@pytest.mark.parametrize("name", ["carlike_min_time_n50", "carlike_min_time_n20", "unicycle_quadratic_n20", "bicycle_min_time_n30"])
def test_gpu_reproduces_golden_fixtures(m, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, _ = _cases(m)[name]
    s = m.BatchSolver(cfg, max_batch=g["x0"].shape[0])
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6
    assert np.abs(r.u - g["u"]).max() < 1e-6
    assert np.abs(r.dt - g["dt"]).max() < 1e-8
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all() and np.median(np.abs(r.iters - g["iters"])) == 0
    s.close()


@pytest.mark.parametrize("name", ["carlike_min_time_n20_monotone", "unicycle_quadratic_n20_monotone"])
def test_gpu_reproduces_the_fixtures_of_the_other_barrier_rule(m, name):
    """tests/golden/*_monotone.npz are made with mu_strategy = monotone (Fiacco-McCormick) instead of the adaptive rule of every other fixture: the device under
    MPC_MU_MONOTONE reproduces them at 1e-6 with the oracle's iteration counts; the device's DEFAULT (adaptive) solves end at the same minimum within the scatter
    tests/test_oracle_solver.py::test_answers_under_the_other_barrier_rule documents (states within 1e-4, same travel time)."""
    from mpc_local_planner_amd import _abi as A
    g = np.load(os.path.join(GOLD, name + ".npz"))
    mk = m.config_carlike_min_time if name.startswith("carlike") else m.config_unicycle_quadratic
    B = g["x0"].shape[0]
    s = m.BatchSolver(mk(20, mu_strategy=A.MU_MONOTONE), max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"]); s.close()
    assert (r.status == 0).all() and np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6 and np.abs(r.dt - g["dt"]).max() < 1e-8
    assert np.abs(r.iters - g["iters"]).max() <= 2
    s = m.BatchSolver(mk(20), max_batch=B)
    a = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"]); s.close()
    assert (a.status == 0).all() and np.abs(a.x - g["x"]).max() < 1e-4 and np.abs(a.u - g["u"]).max() < 3e-4
    if name.startswith("carlike"):
        assert np.abs(a.dt - g["dt"]).max() < 1e-7 * np.abs(g["dt"]).max()


@pytest.mark.parametrize("name", ["carlike_min_time_n20_merit", "carlike_min_time_n50_merit", "bicycle_min_time_n30_merit"])
def test_gpu_reproduces_the_fixtures_of_the_other_line_search(m, name):
    """tests/golden/*_merit.npz are made with the l1-merit line search (the globalisation of rounds 1-5) instead of the filter line search of every other fixture: the device under
    MPC_LS_MERIT reproduces them at 1e-6 with the oracle's iteration counts (tests/test_oracle_solver.py::test_answers_under_the_other_line_search: the CPU half)."""
    from mpc_local_planner_amd import _abi as A
    g = np.load(os.path.join(GOLD, name + ".npz"))
    mk, n = {"carlike_min_time_n20_merit": (m.config_carlike_min_time, 20), "carlike_min_time_n50_merit": (m.config_carlike_min_time, 50),
             "bicycle_min_time_n30_merit": (m.config_bicycle_min_time, 30)}[name]
    B = g["x0"].shape[0]
    s = m.BatchSolver(mk(n, line_search=A.LS_MERIT), max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"]); s.close()
    assert (r.status == 0).all() and np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6 and np.abs(r.dt - g["dt"]).max() < 1e-8
    assert np.abs(r.iters - g["iters"]).max() <= 2 and np.median(np.abs(r.iters - g["iters"])) == 0


def test_config5_shape_vs_slsqp_on_the_device(m):
    """tests/golden/cold_start_scipy_config5.npz: scipy SLSQP on the reference-form NLP of the config-5 shape (bicycle, n = 120), from the reference's cold start.  The device's
    reference path ends at SLSQP's point on at least 4 of the 6 instances SLSQP solves (5 under the l1-merit line search; headings modulo 2 pi;
    tests/test_oracle_solver.py::test_config5_shape_vs_slsqp has the same comparison for the C oracle)."""
    g = np.load(os.path.join(GOLD, "cold_start_scipy_config5.npz"))
    K = int(g["count"])
    from mpc_local_planner_amd import _abi as A
    for ls, floor in ((A.LS_DEFAULT, 4), (A.LS_MERIT, 5)):
        s = m.BatchSolver(m.config_bicycle_min_time(120, line_search=ls), max_batch=K)
        r = s.solve(*m.workloads.bicycle_min_time_inputs(K)); s.close()
        d = r.x - g["x"]
        d[..., 2] = (d[..., 2] + np.pi) % (2 * np.pi) - np.pi
        same = (np.abs(d).reshape(K, -1).max(1) < 1e-5) & (r.status == 0) & g["success"].astype(bool)
        assert same.sum() >= floor and (r.status == 0).sum() >= 7
        assert np.abs(119 * r.dt[same] - g["objective"][same]).max() < 1e-5


def _feasibility(R, ocfg, x0, xf, up, dtp, res, i):
    inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]))
    nlp = R.ReferenceNlp(ocfg, inp)
    z = nlp.pack(R.Trajectory(res.x[i], res.u[i, :-1], float(res.dt[i])))
    lb, ub = nlp.bounds()
    return max(np.abs(nlp.equalities(z)).max(), nlp.inequalities(z).max(initial=0.0), (lb - z).max(), (z - ub).max())


from _parity import account as _account      # noqa: E402  (tests/_parity.py)


def test_config2_batch_vs_c_oracle(m, c_oracle):
    """BASELINE.json config 2 (car-like min-time, n=50) on the SURVEY 8d input distribution."""
    from oracle import se2_nlp as R
    B = 256
    cfg, ocfg = _cases(m)["carlike_min_time_n50"]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    oc = c_oracle.from_nlp_config(ocfg)
    xo, uo, do, st, it = c_oracle.solve_batch(oc, x0, xf, up, dtp)
    both = (r.status == 0) & (st == 0)
    assert both.mean() > 0.85
    assert (r.status == st).mean() > 0.95
    err = np.maximum(np.abs(r.x - xo).reshape(B, -1).max(1), np.abs(r.u - uo).reshape(B, -1).max(1))
    err = np.maximum(err, np.abs(r.dt - do))
    # same algorithm, independent linear algebra (Riccati sweep vs banded LU): identical to round-off except where a
    # line-search/regularisation decision flips on a tie and the iterate sequences part ways
    assert np.median(err[both]) < 1e-8
    _account("config 2, B=256", ocfg, (x0, xf, up, dtp), r, (xo, uo, do, st, it))
    # every converged GPU result is a feasible point of the REFERENCE-form NLP (independent of any solver)
    for i in np.nonzero(r.status == 0)[0][:64]:
        assert _feasibility(R, ocfg, x0, xf, up, dtp, r, i) < 1e-6
    s.close()


def test_config2_batch_under_the_l1_merit_vs_c_oracle(m, c_oracle):
    """mpc_config.line_search = MPC_LS_MERIT (the globalisation of rounds 1-5, kept as an option) against the C oracle with oracle_config.line_search = 0, on config 2: same
    statuses, same iteration counts, same trajectories -- and a different iteration from the default (the filter line search) on part of the batch."""
    from mpc_local_planner_amd import _abi as A
    B = 256
    _, ocfg = _cases(m)["carlike_min_time_n50"]
    inputs = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(m.config_carlike_min_time(50, line_search=A.LS_MERIT), max_batch=B)
    r = s.solve(*inputs); s.close()
    s = m.BatchSolver(m.config_carlike_min_time(50, line_search=A.LS_FILTER), max_batch=B)
    f = s.solve(*inputs); s.close()
    s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=B)
    d = s.solve(*inputs); s.close()
    assert np.array_equal(f.x, d.x) and np.array_equal(f.iters, d.iters) and np.array_equal(f.status, d.status)        # MPC_LS_DEFAULT is the filter
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, line_search=0), *inputs)
    both = (r.status == 0) & (st == 0)
    err = np.abs(r.x - xo).reshape(B, -1).max(1)
    assert (r.status == st).mean() > 0.97 and (r.iters == it)[both].mean() > 0.95 and np.median(err[both]) < 1e-8
    assert (r.iters != f.iters).mean() > 0.2
    print(f"[config 2, B=256] l1 merit: converged {int((r.status == 0).sum())}, mean iterations {r.iters.mean():.2f}; filter: converged {int((f.status == 0).sum())}, mean iterations {f.iters.mean():.2f}")


def test_config2_full_batch_accounting(m, c_oracle):
    """BASELINE.json config 2 at its FULL batch (B = 1024, single candidate = the reference path) against the C oracle, instance by
    instance: match / other KKT point of the reference-form NLP / unclassified (must be 0)."""
    B = 1024
    cfg, ocfg = _cases(m)["carlike_min_time_n50"]
    inputs = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(*inputs)
    out = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *inputs)
    match, other = _account("config 2, B=1024", ocfg, inputs, r, out)
    assert (r.status == out[3]).mean() > 0.97 and match.sum() > 0.85 * B
    s.close()


def test_candidates_full_batch_winners_vs_oracle_rule(m, c_oracle):
    """Candidate initial trajectories on the device (seeds generated in the kernel, hedged workgroups, winner picked with atomics) against
    the identical rule run on the C oracle (oracle/candidates.py), config 2 at B = 1024: converged fraction >= 0.99, winner's iterations
    <= the cap, winners compared instance by instance; every instance whose winner or trajectory differs must still be a KKT point of
    the reference-form NLP."""
    from oracle import candidates as OC
    from mpc_local_planner_amd import _abi as A
    B, n = 1024, 50
    kinds, caps, pars = (A.CAND_REFERENCE, A.CAND_HERMITE_FF, A.CAND_HERMITE_FF, A.CAND_HERMITE_FR), (60, 45, 40, 35), (0.0, 2.0, 3.0, 1.5)      # bench.py's set
    _, ocfg = _cases(m)["carlike_min_time_n50"]
    inputs = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(m.config_carlike_min_time(n, candidates=kinds, candidate_max_iter=caps, candidate_param=pars), max_batch=B)
    r = s.solve(*inputs)
    win, tot = s.last_candidates(B)
    r2 = s.solve(*inputs)                                   # hedging is timing dependent, the RESULT must not be
    win2, _ = s.last_candidates(B)
    np.testing.assert_array_equal(r.x, r2.x); np.testing.assert_array_equal(win, win2); np.testing.assert_array_equal(r.iters, r2.iters)
    ox, ou, od, ost, oit, owin, olow, allr = OC.solve_candidates(c_oracle, lambda cap: c_oracle.from_nlp_config(ocfg, max_iter=cap), *inputs, kinds, caps, n, ocfg.dt_ref, params=pars)
    conv = r.status == 0
    print(f"[candidates] device converged {conv.mean():.4f} (oracle rule {np.mean(ost == 0):.4f}); winners device {np.bincount(win + 1, minlength=5).tolist()} "
          f"oracle {np.bincount(owin + 1, minlength=5).tolist()} (index 0 = none); equal winners {np.mean(win == owin):.4f}; winner iterations p99 {np.percentile(r.iters[conv], 99):.0f}")
    assert conv.mean() >= 0.99 and (win[conv] >= 0).all() and (win[~conv] == -1).all()
    assert (r.iters[conv] <= np.asarray(caps)[win[conv]]).all() and (tot >= r.iters).all()
    assert np.mean(win == owin) > 0.97
    # the single-candidate solve IS candidate 0: where it converges within the cap it must win and supply the identical trajectory
    s1 = m.BatchSolver(m.config_carlike_min_time(n, max_iter=caps[0]), max_batch=B)
    r1 = s1.solve(*inputs)
    ref_ok = r1.status == 0
    assert (win[ref_ok] == 0).all() and np.array_equal(r.x[ref_ok], r1.x[ref_ok]) and np.array_equal(r.iters[ref_ok], r1.iters[ref_ok])
    assert (win[~ref_ok] != 0).all()
    s1.close()
    # what the hedges cost in solution quality: the instances a hedge answers although the reference path alone -- one candidate, the
    # reference's budget of 100 iterations -- solves them too: travel time (n - 1) dt of the hedge's answer minus that of the reference path's answer
    s100 = m.BatchSolver(m.config_carlike_min_time(n, max_iter=100), max_batch=B)
    r100 = s100.solve(*inputs)
    s100.close()
    sel = conv & (win > 0) & (r100.status == 0)
    dobj = (n - 1) * (r.dt[sel] - r100.dt[sel])
    print(f"[hedges vs the reference path alone] {int(sel.sum())} instances answered by a hedge although the 100-iteration reference path solves them: objective difference "
          f"median {np.median(dobj):+.3f} s, p10 {np.percentile(dobj, 10):+.3f}, p90 {np.percentile(dobj, 90):+.3f}, max {dobj.max():+.3f}; better {np.mean(dobj < -1e-6):.2f} same {np.mean(np.abs(dobj) < 1e-6):.2f} worse {np.mean(dobj > 1e-6):.2f}")
    assert sel.sum() >= 30 and np.median(dobj) <= 1e-9 and np.mean(dobj > 1e-6) < 0.15          # r04 with the inertia test: 43 such instances (100 before: the reference path needs fewer iterations)
    _account("config 2 with candidates, B=1024", ocfg, inputs, r, (ox, ou, od, ost, oit))
    s.close()


def test_full_size_properties_config2(m):
    """B = 1024 (BASELINE.json config 2): determinism, batch-permutation equivariance, SE(2) equivariance."""
    B = 1024
    cfg = m.config_carlike_min_time(50)
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(cfg, max_batch=B)
    a = s.solve(x0, xf, up, dtp)
    b = s.solve(x0, xf, up, dtp)
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_array_equal(a.iters, b.iters)
    perm = np.random.default_rng(0).permutation(B)
    c = s.solve(x0[perm], xf[perm], up[perm], dtp[perm])
    np.testing.assert_array_equal(c.x, a.x[perm])
    np.testing.assert_array_equal(c.status, a.status[perm])
    assert (a.status == 0).mean() > 0.85
    ok = a.status == 0
    # bounds and time-series conventions on every converged instance
    assert a.u[ok, :, 0].max() <= 0.4 + 1e-9 and a.u[ok, :, 0].min() >= -0.2 - 1e-9
    assert np.abs(a.u[ok, :, 1]).max() <= 1.4 + 1e-9
    assert (a.dt[ok] > 0).all() and (a.dt[ok] < 10).all()
    np.testing.assert_array_equal(a.u[:, -1], a.u[:, -2])
    np.testing.assert_array_equal(a.x[:, 0], x0)
    np.testing.assert_array_equal(a.x[:, -1, :2], xf[:, :2])
    # rigid-motion equivariance: rotate + translate the whole problem
    phi, t = 0.7, np.array([3.0, -2.0])
    Rm = np.array([[np.cos(phi), -np.sin(phi)], [np.sin(phi), np.cos(phi)]])
    def move(p):
        q = p.copy()
        q[..., :2] = p[..., :2] @ Rm.T + t
        q[..., 2] = (p[..., 2] + phi + np.pi) % (2 * np.pi) - np.pi
        return q
    d = s.solve(move(x0), move(xf), up, dtp)
    both = ok & (d.status == 0)
    assert both.mean() > 0.8
    ex = np.abs(d.x[..., :2] - move(a.x)[..., :2]).reshape(B, -1).max(1)
    assert np.median(ex[both]) < 1e-7
    assert (ex[both] < 1e-4).mean() > 0.85
    s.close()


def test_device_pointer_entry_and_kernel_timer(m):
    import torch
    B, n = 128, 50
    cfg = m.config_carlike_min_time(n)
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(cfg, max_batch=B)
    host = s.solve(x0, xf, up, dtp)
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(a).to(dev)
    dx0, dxf, dup, ddtp = T(x0), T(xf), T(up), T(dtp)
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev)
    uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    s.solve_device(B, dx0.data_ptr(), dxf.data_ptr(), dup.data_ptr(), ddtp.data_ptr(), None, None, None,
                   xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
    s.synchronize()
    assert s.last_kernel_ms() > 0
    np.testing.assert_array_equal(xo.cpu().numpy(), host.x)
    np.testing.assert_array_equal(st.cpu().numpy(), host.status)
    s.close()


def test_edge_cases(m):
    from mpc_local_planner_amd._abi import MPC_EBATCH
    cfg = m.config_carlike_min_time(20)
    s = m.BatchSolver(cfg, max_batch=70)
    # B = 1 and a ragged batch (not a multiple of the wavefront size)
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(70, seed=3, goal_range=(1.0, 2.5))
    full = s.solve(x0, xf, up, dtp)
    one = s.solve(x0[:1], xf[:1], up[:1], dtp[:1])
    np.testing.assert_array_equal(one.x[0], full.x[0])
    # dt_prev = 0 (first cycle): stage-0 rate rows dropped (stage_inequality_se2.cpp:197-201)
    z = s.solve(x0[:8], xf[:8], np.zeros((8, 2)), np.zeros(8))
    assert (z.status == 0).sum() >= 6
    # optional inputs omitted entirely
    z2 = s.solve(x0[:8], xf[:8])
    np.testing.assert_array_equal(z2.x, z.x)
    # warm start path: vertex values handed in; x_0 and the fixed goal are overwritten by the inputs
    w = s.solve(x0[:8], xf[:8], up[:8], dtp[:8], init=(full.x[:8], full.u[:8], full.dt[:8]))
    np.testing.assert_array_equal(w.x[:, 0], x0[:8])
    np.testing.assert_array_equal(w.x[:, -1], xf[:8])
    # capacity error, no crash
    with pytest.raises(m.MpcError) as ei:
        s.solve(np.zeros((71, 3)), np.ones((71, 3)))
    assert ei.value.code == MPC_EBATCH
    s.close()
    # smallest grid
    s3 = m.BatchSolver(m.config_carlike_min_time(3), max_batch=2)
    r3 = s3.solve(np.array([[0, 0, 0.0]]), np.array([[0.3, 0, 0.0]]))
    assert r3.status[0] == 0 and r3.dt[0] > 0
    s3.close()


def test_fp32_path_and_other_models(m, c_oracle):
    from oracle import se2_nlp as R
    n = 30
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(64, seed=21, goal_range=(2.0, 6.0))
    s64 = m.BatchSolver(m.config_bicycle_min_time(n), max_batch=64)
    s32 = m.BatchSolver(m.config_bicycle_min_time(n, precision=1, tol=1e-4), max_batch=64)
    a, b = s64.solve(x0, xf, up, dtp), s32.solve(x0, xf, up, dtp)
    both = (a.status == 0) & (b.status == 0)
    assert both.sum() >= 32
    err = np.abs(a.x - b.x).reshape(64, -1).max(1)
    # what PLAIN fp32 achieves against the fp64 solve of the same instances (BASELINE configs[4] names fp32; the north-star tolerance is 1e-4): the median
    # sits between 1e-5 and 5e-3, i.e. fp32 alone does NOT meet 1e-4 for most instances -- MPC_MIXED is the precision that does
    # (test_mixed_precision_meets_the_fp64_tolerance_on_config5, and the accounting below at B = 1024); bench.py reports both legs with `meets_1e-4`
    med, frac = float(np.median(err[both])), float((err[both] < 1e-4).mean())
    print(f"plain fp32 vs fp64, bicycle n = {n}: median |dx| {med:.1e}, within 1e-4: {frac:.2f} of {int(both.sum())}")
    assert 1e-6 < med < 5e-3
    assert frac < 0.9                                    # if this ever fails, fp32 has become good enough and the bench legs should say so
    oc = c_oracle.from_nlp_config(R.config_bicycle_min_time(n))
    xo, uo, do, st, it = c_oracle.solve_batch(oc, x0, xf, up, dtp)
    bo = (a.status == 0) & (st == 0)
    assert np.median(np.abs(a.x - xo).reshape(64, -1).max(1)[bo]) < 1e-8
    s64.close(); s32.close()
    # front-wheel car
    fw = m.make_config(model=2, model_params=(0.4,), n=20, u_lb=(-0.2, -1.0), u_ub=(0.4, 1.0), du_lb=(-0.5, -0.5), du_ub=(0.5, 0.5))
    sf = m.BatchSolver(fw, max_batch=16)
    rf = sf.solve(x0[:16] * 0.3, xf[:16] * 0.3, up[:16], dtp[:16])
    assert (rf.status == 0).sum() >= 10
    sf.close()


def test_obstacle_rows_golden_and_properties(m):
    """Clearance rows (stage_inequality_se2.cpp:50-175): association on the device + d_min - dist <= 0 rows.
    (i) golden fixtures of the numpy oracle (same association, same iterate sequence); (ii) on a config-3-shaped batch
    (n=80, 16 convex polygons) every converged trajectory keeps every ASSOCIATED obstacle at >= d_min."""
    from oracle import se2_nlp as R
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_obstacles_n30.npz"))
    B, O, V = g["x0"].shape[0], g["vertices"].shape[1], g["vertices"].shape[2]
    cfg = m.config_unicycle_quadratic(30, max_obstacles=O, max_vertices=V, max_obstacle_rows=int(g["max_rows"]))
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"]))
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all()
    # obstacles are required once the solver was created for them
    with pytest.raises(m.MpcError):
        s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    s.close()
    # config 3 shape
    n, B, O, V, M = 80, 256, 16, 6, 4
    x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
    cfg = m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M)
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
    ok = r.status == 0
    assert ok.mean() > 0.95
    ocfg = R.config_unicycle_quadratic(n)
    # the same batch on the C oracle (banded LU, clearance rows condensed the same way): same convergence set, same solutions
    from oracle import c_oracle as CO
    oc = CO.from_nlp_config(ocfg)
    xo, uo, do, st, it = CO.solve_batch(oc, x0, xf, up, dtp, obstacles=(no, nv, verts), obst=CO.obst_from_nlp_config(ocfg, O, V, M))
    both = ok & (st == 0)
    assert both.sum() >= 0.95 * B and abs(int(ok.sum()) - int((st == 0).sum())) <= 3
    err = np.abs(r.x - xo).reshape(B, -1).max(1)
    assert np.median(err[both]) < 1e-8
    assert np.median(np.abs(r.iters[both] - it[both])) <= 1
    _account("config 3 shape, rows inactive placement, B=256", ocfg, (x0, xf, up, dtp), r, (xo, uo, do, st, it), obstacles=(no, nv, verts), max_rows=M, min_match=0.9)
    checked = 0
    for i in np.nonzero(ok)[0][:24]:
        obs = [R.Obstacle(R.OBST_POLYGON, verts[i, o, :nv[i, o]]) for o in range(no[i])]
        rel, _ = R.associate_obstacles(ocfg, R.cold_start(ocfg, x0[i], xf[i]), obs, max_rows=M)
        for k in range(1, n - 1):
            for j in rel[k]:
                assert R.footprint_distance(R.FOOTPRINT_POINT, (), r.x[i, k], obs[j]) >= cfg.min_obstacle_dist - 1e-6
                checked += 1
        # and the dynamics / boxes still hold
        nlp = R.ReferenceNlp(ocfg, R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i])))
        z = nlp.pack(R.Trajectory(r.x[i], r.u[i, :-1], float(r.dt[i])))
        assert np.abs(nlp.equalities(z)).max() < 1e-6
    assert checked > 500
    s.close()


def test_cpp_controller_facade_closed_loop(m, tmp_path):
    """Builds tests/gpu_controller_demo.cpp against include/mpc_controller.hpp + libmpc_hip.so and runs the reference's
    stand-alone scenario (src/test_mpc_optim_node.cpp) in closed loop through the C++ Controller facade."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "mpc_local_planner_amd", "csrc")
    exe = str(tmp_path / "controller_demo")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(root, "tests", "gpu_controller_demo.cpp"), "-L" + libdir, "-lmpc_hip",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "DEMO_OK" in r.stdout, r.stdout + r.stderr


def test_two_handles_shard_a_batch_like_two_gpus(m, tmp_path):
    """examples/two_handles_shard.cpp: the multi-GPU split from a C++ host through the C ABI alone -- one handle per device (device 1 when the box has one, else both on
    device 0), one host thread per handle, contiguous ragged shards of one batch with hedged candidates -- returns bit for bit what ONE handle returns for the
    whole batch.  No scaling curve is claimed from it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "mpc_local_planner_amd", "csrc")
    exe = str(tmp_path / "two_handles_shard")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(root, "examples", "two_handles_shard.cpp"), "-L" + libdir, "-lmpc_hip",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "SHARD_OK" in r.stdout, r.stdout + r.stderr


def test_no_uninitialised_lds_reads(m, tmp_path):
    """Instrumented build (-DMPC_POISON_LDS fills the whole LDS working set with NaN before the solve): every golden
    fixture must still be reproduced, i.e. the solver never consumes an LDS word it has not written (regression test
    for a 0 * garbage = NaN bug that only showed up when a previous kernel had left NaN patterns in LDS)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libmpc_hip_poison.so")
    from mpc_local_planner_amd import _lib
    _lib.build(extra_flags=["-DMPC_POISON_LDS"], out=so)          # the split build of the product library with one more flag (parallel: ~25 s)
    # round-2 paths: candidates (kernel-generated seeds), kept multipliers, a turning footprint against polygons and a dynamic obstacle -- results of
    # the normal library (this process) must be reproduced bit for bit by the poisoned build (the subprocess)
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(16, seed=77, goal_range=(2.0, 4.0))
    ckw = dict(candidates=(0, 5, 3, 7), candidate_max_iter=(40, 40, 40, 40), candidate_param=(0, 2.0, 0, 1.5), dual_warm_start=True)
    sA = m.BatchSolver(m.config_carlike_min_time(30, **ckw), max_batch=16)
    rA = sA.solve(x0, xf, up, dtp)
    rA2 = sA.solve(x0 + 0.01, xf, up, dtp, init=(rA.x, rA.u, rA.dt))
    sA.close()
    d = xf[:, :2] - x0[:, :2]
    nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    vt = np.zeros((16, 2, 4, 2)); nv = np.zeros((16, 2), np.int32); rad = np.zeros((16, 2)); vel = np.zeros((16, 2, 2))
    sq = np.array([[0.2, 0.2], [-0.2, 0.2], [-0.2, -0.2], [0.2, -0.2]])
    vt[:, 0] = (x0[:, None, :2] + 0.5 * d[:, None, :] + 0.9 * nrm[:, None, :]) + sq[None]; nv[:, 0] = 4
    vt[:, 1, 0] = x0[:, :2] + 0.6 * d - 1.1 * nrm; nv[:, 1] = 1; rad[:, 1] = 0.1; vel[:, 1] = 0.1 * nrm
    okw = dict(footprint_kind=2, footprint_params=(0.0, 0.0, 0.4, 0.0), min_obstacle_dist=0.25, enable_dynamic_obstacles=True, max_obstacles=2, max_vertices=4, max_obstacle_rows=4)
    sB = m.BatchSolver(m.config_carlike_min_time(30, **okw), max_batch=16)
    rB = sB.solve(x0, xf, up, dtp, obstacles=(np.full(16, 2, np.int32), nv, vt, rad, vel))
    sB.close()
    np.savez(str(tmp_path / "r2.npz"), x0=x0, xf=xf, up=up, dtp=dtp, xA=rA.x, stA=rA.status, xA2=rA2.x, itA2=rA2.iters, nv=nv, vt=vt, rad=rad, vel=vel, xB=rB.x, stB=rB.status)
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import mpc_local_planner_amd as m
R2 = %r
if R2:
    g = np.load(R2)
    ckw = dict(candidates=(0, 5, 3, 7), candidate_max_iter=(40, 40, 40, 40), candidate_param=(0, 2.0, 0, 1.5), dual_warm_start=True)
    s = m.BatchSolver(m.config_carlike_min_time(30, **ckw), max_batch=16)
    r = s.solve(g["x0"], g["xf"], g["up"], g["dtp"])
    assert np.array_equal(r.x, g["xA"]) and np.array_equal(r.status, g["stA"]), "candidates"
    r2 = s.solve(g["x0"] + 0.01, g["xf"], g["up"], g["dtp"], init=(r.x, r.u, r.dt))
    assert np.array_equal(r2.x, g["xA2"]) and np.array_equal(r2.iters, g["itA2"]), "kept multipliers"
    s.close()
    okw = dict(footprint_kind=2, footprint_params=(0.0, 0.0, 0.4, 0.0), min_obstacle_dist=0.25, enable_dynamic_obstacles=True, max_obstacles=2, max_vertices=4, max_obstacle_rows=4)
    s = m.BatchSolver(m.config_carlike_min_time(30, **okw), max_batch=16)
    r = s.solve(g["x0"], g["xf"], g["up"], g["dtp"], obstacles=(np.full(16, 2, np.int32), g["nv"], g["vt"], g["rad"], g["vel"]))
    assert np.array_equal(r.x, g["xB"]) and np.array_equal(r.status, g["stB"]), "turning footprint + polygon + dynamic obstacle"
    s.close()
cases = {"carlike_min_time_n50": m.config_carlike_min_time(50), "unicycle_quadratic_n20": m.config_unicycle_quadratic(20),
         "bicycle_min_time_n30": m.config_bicycle_min_time(30)}
for name, cfg in cases.items():
    g = np.load(%r + "/" + name + ".npz")
    s = m.BatchSolver(cfg, max_batch=8)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all() and np.abs(r.x - g["x"]).max() < 1e-6, (name, r.status, r.iters)
g = np.load(%r + "/unicycle_quadratic_obstacles_n30.npz")
O, V = g["vertices"].shape[1], g["vertices"].shape[2]
s = m.BatchSolver(m.config_unicycle_quadratic(30, max_obstacles=O, max_vertices=V, max_obstacle_rows=int(g["max_rows"])), max_batch=8)
r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"]))
assert (r.status == 0).all() and np.abs(r.x - g["x"]).max() < 1e-6, ("obstacles", r.status)
print("POISON_OK")
""" % (root, str(tmp_path / "r2.npz"), GOLD, GOLD)
    env = dict(os.environ, MPC_HIP_LIB=so)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert "POISON_OK" in r.stdout, r.stdout + r.stderr


def test_warm_start_golden(m):
    """Second control cycle: previous solution as the initial guess, x0 advanced by the plant (numpy-oracle fixture
    tests/golden/carlike_min_time_n20_warm.npz, generated by make_golden.py --warm)."""
    g = np.load(os.path.join(GOLD, "carlike_min_time_n20_warm.npz"))
    s = m.BatchSolver(m.config_carlike_min_time(20), max_batch=g["x0"].shape[0])
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], init=(g["x_init"], g["u_init"], g["dt_init"]))
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6 and np.abs(r.dt - g["dt"]).max() < 1e-8
    assert (np.abs(r.iters - g["iters"]) <= 2).all()
    s.close()


def test_mixed_grid_sizes_in_one_batch(m):
    """mpc_set_grid_sizes: instances with n_i = 50 and n_i = 20 grid points in ONE launch of a solver created for n = 50
    (grid adaptation changes n between cycles, finite_differences_variable_grid_se2.cpp:99-121); each instance must land
    on its own golden solution; rows beyond n_i of the outputs repeat the last grid point."""
    g50 = np.load(os.path.join(GOLD, "carlike_min_time_n50.npz"))
    g20 = np.load(os.path.join(GOLD, "carlike_min_time_n20.npz"))
    b50, b20 = g50["x0"].shape[0], g20["x0"].shape[0]
    cat = lambda k: np.concatenate([g50[k], g20[k]])
    s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=b50 + b20)
    s.set_grid_sizes(np.array([50] * b50 + [20] * b20, dtype=np.int32))
    r = s.solve(cat("x0"), cat("xf"), cat("u_prev"), cat("dt_prev"))
    assert (r.status == 0).all()
    assert np.abs(r.x[:b50] - g50["x"]).max() < 1e-6 and np.abs(r.dt[:b50] - g50["dt"]).max() < 1e-8
    assert np.abs(r.x[b50:, :20] - g20["x"]).max() < 1e-6 and np.abs(r.u[b50:, :19] - g20["u"][:, :19]).max() < 1e-6
    assert np.abs(r.dt[b50:] - g20["dt"]).max() < 1e-8
    assert np.abs(r.x[b50:, 20:] - r.x[b50:, 19:20]).max() == 0.0
    s.set_grid_sizes(None)
    r2 = s.solve(g50["x0"], g50["xf"], g50["u_prev"], g50["dt_prev"])
    assert np.abs(r2.x - g50["x"]).max() < 1e-6
    s.close()


def test_lds_working_set_and_workgroups_per_cu_of_the_baseline_configs(m):
    """mpc_lds_bytes: the working set of one instance in LDS = the dynamic LDS of its workgroup; what the 160 KB of a compute unit hold decides how many wavefronts a
    CU runs (four at most: the register file holds the kernel at one wave per SIMD).  Everything in LDS (MPC_STAGE_LDS, rounds 1-4): 4 at BASELINE configs[1] / [3]
    (n = 50, fp64), 2 at configs[2] until r05 and 1 since (n = 80, 16 polygons of 6 vertices, four clearance rows per grid point), 3 in plain fp32 at configs[4] (n = 120), 1 in fp64 there.
    r05, MPC_STAGE_AUTO: with the factorisation data in global memory where the LDS form leaves half of the SIMDs empty, 4 at configs[2] and 4 at n = 120 in fp64;
    the headline grid, fp32 and the refinement phase of MPC_MIXED keep the LDS form."""
    from mpc_local_planner_amd import _abi as A
    CU = 160 * 1024
    def wgs(cfg):
        s = m.BatchSolver(cfg, max_batch=4)
        b = s.lds_bytes()
        s.close()
        return min(4, CU // b), b
    lds = dict(stage_data=A.STAGE_LDS)
    assert wgs(m.config_carlike_min_time(50))[0] == 4 and wgs(m.config_carlike_min_time(50))[1] == wgs(m.config_carlike_min_time(50, **lds))[1]
    w3, b3 = wgs(m.config_unicycle_quadratic(80, max_obstacles=16, max_vertices=6, max_obstacle_rows=4, **lds))
    assert w3 == 1 and b3 < 84 * 1024, b3          # (2 until the edge table of the polygon distances -- 2.3 KB -- went in: 81 456 B were 464 B short of the limit for two)
    w3, b3 = wgs(m.config_unicycle_quadratic(80, max_obstacles=16, max_vertices=6, max_obstacle_rows=4))
    assert w3 == 4, b3
    assert wgs(m.config_unicycle_quadratic(80, **lds))[0] == 2 and wgs(m.config_unicycle_quadratic(80))[0] == 4
    assert wgs(m.config_bicycle_min_time(120, precision=1, tol=1e-4, **lds))[0] == 3 and wgs(m.config_bicycle_min_time(120, precision=1, tol=1e-4))[0] == 4      # plain fp32: the global form from three per CU on
    assert wgs(m.config_bicycle_min_time(120, **lds))[0] == 1 and wgs(m.config_bicycle_min_time(120))[0] == 4
    assert wgs(m.config_bicycle_min_time(120, precision=A.MIXED))[0] == 1          # the short refinement phase measured faster in the LDS form
    # grids beyond the LDS form's ~215 points exist in the global form
    assert wgs(m.config_bicycle_min_time(300))[0] == 1
    with pytest.raises(Exception):
        m.BatchSolver(m.config_bicycle_min_time(300, **lds), max_batch=4)


def test_fixed_layout_kernel_equals_the_generic_kernel_bit_for_bit(m):
    """The fp64 headline kernel has an instantiation whose LDS layout is a compile-time constant for records of 50 grid points (mpc_wave.hpp::FixedLayout,
    picked by launch_solve when the handle's layout matches).  Same code, same arithmetic: a handle created for n = 51 with every instance set to 50 grid
    points runs the GENERIC kernel on the same problems and must return the same trajectories, controls, dt, statuses and iteration counts bit for bit --
    with one candidate and with the hedged candidates of the headline run."""
    B = 1024
    inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
    for kw in (dict(), dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5))):
        a = m.BatchSolver(m.config_carlike_min_time(50, **kw), max_batch=B)
        ra = a.solve(*inp)
        a.close()
        b = m.BatchSolver(m.config_carlike_min_time(51, **kw), max_batch=B)
        b.set_grid_sizes(np.full(B, 50, dtype=np.int32))
        rb = b.solve(*inp)
        b.close()
        assert (ra.status == 0).mean() > 0.9
        assert np.array_equal(ra.status, rb.status) and np.array_equal(ra.iters, rb.iters) and np.array_equal(ra.dt, rb.dt)
        assert np.array_equal(ra.x, rb.x[:, :50]) and np.array_equal(ra.u[:, :49], rb.u[:, :49])


def test_global_form_equals_lds_form_over_the_grid_sizes(m):
    """The global block's rows have a pitch of (n / 16 + 2) x 16 words, the sweeps' prefetches run into the spare columns, the partitioned sweeps start at 40 grid points and a
    ragged batch uses less than a row: fp64 results of the two forms bit for bit over grid sizes on either side of every such boundary (scripts/dev/gs_sweep.py is the longer list)."""
    from mpc_local_planner_amd import _abi as A
    B = 32
    def both(mk, inp, n_grid=None, **kw):
        out = []
        for mode in (A.STAGE_LDS, A.STAGE_GLOBAL):
            s = m.BatchSolver(mk(mode), max_batch=B)
            if n_grid is not None: s.set_grid_sizes(n_grid)
            out.append(s.solve(*inp, **kw)); s.close()
        for f in ("x", "u", "dt", "status", "iters"):
            assert np.array_equal(getattr(out[0], f), getattr(out[1], f), equal_nan=True), f
        assert (out[0].status == 0).mean() > 0.7
    for n in (12, 39, 40, 48, 49, 64, 65, 128, 129, 200):
        both(lambda mode: m.config_carlike_min_time(n, stage_data=mode), m.workloads.carlike_min_time_inputs(B, seed=n))
    for n in (30, 64, 100):
        x0, xf, up, dtp, obstacles = m.workloads.unicycle_obstacle_inputs(B, n_obst=8, max_vertices=5, lateral=(0.15, 0.8))
        both(lambda mode: m.config_unicycle_quadratic(n, max_obstacles=8, max_vertices=5, max_obstacle_rows=3, max_iter=60, stage_data=mode), (x0, xf, up, dtp), obstacles=obstacles)
    ng = np.random.default_rng(90).integers(8, 91, B).astype(np.int32)
    both(lambda mode: m.config_bicycle_min_time(90, stage_data=mode), m.workloads.bicycle_min_time_inputs(B), n_grid=ng)


def test_two_waves_per_simd_kernel_equals_the_one_wave_kernels_bit_for_bit(m):
    """mpc_config.two_wave_min_batch: launches of at least that many instances take the kernel variant for two resident waves per SIMD (236 registers, no scratch: every phase of an
    iteration on a lane index of its own, the solve loop's uniform scalars in scalar registers, generic line-search trials, no partitioned sweeps; exists where the LDS record fits
    eight times into a CU: n <= 24 in fp64).  Same arithmetic on the same
    numbers: trajectories, controls, dt, statuses and iteration counts bit for bit those of the one-wave kernels -- every model, with candidates, on a ragged batch; and a
    grid whose record does not fit eight times ignores the setting."""
    B = 192
    def both(mk, inp, n_grid=None):
        out = []
        for w2 in (-1, 1):
            s = m.BatchSolver(mk(two_wave_min_batch=w2), max_batch=B)
            if n_grid is not None: s.set_grid_sizes(n_grid)
            out.append(s.solve(*inp)); s.close()
        for f in ("x", "u", "dt", "status", "iters"):
            assert np.array_equal(getattr(out[0], f), getattr(out[1], f), equal_nan=True), f
        assert (out[0].status == 0).mean() > 0.7
    cand = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5))
    for n in (12, 20, 24):
        both(lambda **k: m.config_carlike_min_time(n, **k), m.workloads.carlike_min_time_inputs(B, seed=n, goal_range=(0.5, 2.5)))
        both(lambda **k: m.config_carlike_min_time(n, **cand, **k), m.workloads.carlike_min_time_inputs(B, seed=100 + n, goal_range=(0.5, 2.5)))
    both(lambda **k: m.config_unicycle_quadratic(20, **k), m.workloads.carlike_min_time_inputs(B, seed=3, goal_range=(0.5, 1.5)))
    both(lambda **k: m.config_bicycle_min_time(24, **k), m.workloads.bicycle_min_time_inputs(B, goal_range=(1.0, 5.0)))
    both(lambda **k: m.config_carlike_min_time(24, **k), m.workloads.carlike_min_time_inputs(B, seed=9, goal_range=(0.5, 2.5)), n_grid=(8 + np.arange(B) % 17).astype(np.int32))
    both(lambda **k: m.config_carlike_min_time(50, **k), m.workloads.carlike_min_time_inputs(B, seed=50))      # 40 KB record: one wave per SIMD whatever the setting
    # a warm-started second cycle from kept multipliers (dual_warm_start: the kernel's other entry into the solve loop) and a per-solve time budget
    inp = m.workloads.carlike_min_time_inputs(B, seed=77, goal_range=(0.5, 2.5))
    out = []
    for w2 in (-1, 1):
        s = m.BatchSolver(m.config_carlike_min_time(20, two_wave_min_batch=w2, mu_init_warm=1e-2, dual_warm_start=True, mu_init_dual=1e-3), max_batch=B)
        r1 = s.solve(*inp)
        x1 = inp[0].copy(); x1[:, :2] += 0.02
        r2 = s.solve(x1, inp[1], r1.u[:, 0].copy(), inp[3], init=(r1.x.copy(), r1.u.copy(), r1.dt.copy()))
        out.append((r1, r2)); s.close()
    for f in ("x", "u", "dt", "status", "iters"):
        assert np.array_equal(getattr(out[0][0], f), getattr(out[1][0], f), equal_nan=True) and np.array_equal(getattr(out[0][1], f), getattr(out[1][1], f), equal_nan=True), f
    assert (out[0][1].status == 0).mean() > 0.7 and out[0][1].iters.mean() < out[0][0].iters.mean()


def test_occupancy_reported_by_the_runtime(m):
    """mpc_occupancy: resident one-wave workgroups per CU of the kernel instantiation a launch of B instances selects, from hipOccupancyMaxActiveBlocksPerMultiprocessor (registers,
    LDS) -- what bench.py reports as workgroups_per_cu / waves_per_simd and what sizes the per-XCD block pools"""
    s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=64)
    assert s.occupancy(64) == (4, s.lds_bytes()) and s.lds_bytes() == 40128          # BASELINE configs[1]: one wave per SIMD, 40 KB record
    s.close()
    s = m.BatchSolver(m.config_carlike_min_time(20), max_batch=8192)
    w_small, lds_small = s.occupancy(1024)
    w_large, lds_large = s.occupancy(8192)
    assert (w_small, w_large) == (4, 8) and lds_small == lds_large == s.lds_bytes()   # the two-waves-per-SIMD kernel from the default threshold on
    assert s.occupancy(4096)[0] == 8 and s.occupancy(4095)[0] == 4                     # ... which is 4096 instances per launch
    s.close()
    s = m.BatchSolver(m.config_carlike_min_time(20, two_wave_min_batch=-1), max_batch=8192)
    assert s.occupancy(8192)[0] == 4
    s.close()
    s = m.BatchSolver(m.config_bicycle_min_time(120), max_batch=64)
    assert s.occupancy(64)[0] == 4 and s.lds_bytes() < 40960                          # BASELINE configs[4] shape in fp64: the global form, four per CU
    s.close()


@pytest.mark.parametrize("case", ["bicycle_n120_fp64_candidates", "bicycle_n120_mixed", "unicycle_n80_polygons", "carlike_n50_candidates", "carlike_n30_ragged_fp32"])
def test_factorisation_data_in_global_memory_equals_lds_bit_for_bit(m, case):
    """mpc_config.stage_data: the stage records and Riccati gains of a solve (63 of the 97 words per grid point) live in LDS or in a per-workgroup block of
    global memory (mpc_wave.hpp::GlobalStage; what MPC_STAGE_AUTO picks when the smaller LDS record puts more workgroups on a CU: n = 120 in fp64 1 -> 4,
    n = 80 with 16 polygons 2 -> 3).  Same arithmetic on the same numbers: trajectories, controls, dt, statuses and iteration counts must be equal bit for bit,
    in every precision, with candidates, with clearance rows, and on ragged grids -- and MPC_STAGE_AUTO must be one of the two."""
    from mpc_local_planner_amd import _abi as A
    obstacles, sizes = None, None
    if case == "bicycle_n120_fp64_candidates":
        B, mk = 512, lambda **k: m.config_bicycle_min_time(120, candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0), **k)
        inp = m.workloads.bicycle_min_time_inputs(B)
    elif case == "bicycle_n120_mixed":
        B, mk = 512, lambda **k: m.config_bicycle_min_time(120, precision=A.MIXED, candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0), **k)
        inp = m.workloads.bicycle_min_time_inputs(B)
    elif case == "unicycle_n80_polygons":
        B, O, V = 512, 16, 6
        x0, xf, up, dtp, obstacles = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=(0.15, 0.8))
        inp = (x0, xf, up, dtp)
        mk = lambda **k: m.config_unicycle_quadratic(80, max_obstacles=O, max_vertices=V, max_obstacle_rows=4, max_iter=60, **k)
    elif case == "carlike_n50_candidates":
        B, mk = 1024, lambda **k: m.config_carlike_min_time(50, candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5), **k)
        inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
    else:
        B, mk = 256, lambda **k: m.config_carlike_min_time(30, precision=A.FP32, tol=1e-4, **k)
        inp = m.workloads.carlike_min_time_inputs(B, seed=5)
        sizes = (12 + np.arange(B) % 19).astype(np.int32)          # grids of 12 .. 30 points in one batch (serial sweeps below 40 points)
    res, lds = {}, {}
    for mode in (A.STAGE_LDS, A.STAGE_GLOBAL, A.STAGE_AUTO):
        s = m.BatchSolver(mk(stage_data=mode), max_batch=B)
        if sizes is not None:
            s.set_grid_sizes(sizes)
        res[mode] = s.solve(*inp, obstacles=obstacles)
        lds[mode] = s.lds_bytes()
        s.close()
    a, g, auto = res[A.STAGE_LDS], res[A.STAGE_GLOBAL], res[A.STAGE_AUTO]
    assert (a.status == 0).mean() > 0.7
    fp32_phase = case in ("bicycle_n120_mixed", "carlike_n30_ragged_fp32")
    for f in ("status", "iters", "dt", "x", "u"):
        # fp64: bit for bit.  An fp32 (phase) in the global form agrees to rounding only: the compiler contracts / packs the fp32 lane-parallel passes differently around
        # global loads (each form is deterministic run to run; MPC_STAGE_AUTO never takes the global form in fp32)
        if not fp32_phase:
            assert np.array_equal(getattr(a, f), getattr(g, f)), f
        assert np.array_equal(getattr(a, f), getattr(auto, f)), f
    if fp32_phase:
        both = (a.status == 0) & (g.status == 0) & (a.iters == g.iters)
        assert abs((a.status == 0).mean() - (g.status == 0).mean()) < 0.02 and both.mean() > 0.5
        assert np.median(np.abs(a.x - g.x).reshape(B, -1).max(1)[both]) < (1e-6 if case == "bicycle_n120_mixed" else 1e-3)
    assert lds[A.STAGE_GLOBAL] < 0.55 * lds[A.STAGE_LDS]
    # what MPC_STAGE_AUTO picks in fp64: the global form where the LDS form leaves at least half of a CU's SIMDs empty and the global form fills more (four workgroups
    # per CU at most: one wave per SIMD); plain fp32 takes the global form from three per CU on (the ragged n = 30 case here has four), the fp32 phase of MPC_MIXED keeps the LDS form.  mpc_lds_bytes reports the fp64 kernel's record (MPC_MIXED: the refinement phase's)
    per_cu = lambda b: min(4, (160 * 1024) // b)
    want_global = case not in ("carlike_n30_ragged_fp32", "bicycle_n120_mixed") and per_cu(lds[A.STAGE_LDS]) <= 2 and per_cu(lds[A.STAGE_GLOBAL]) > per_cu(lds[A.STAGE_LDS])
    assert lds[A.STAGE_AUTO] == (lds[A.STAGE_GLOBAL] if want_global else lds[A.STAGE_LDS])
    if case in ("bicycle_n120_fp64_candidates", "unicycle_n80_polygons"):
        assert lds[A.STAGE_AUTO] == lds[A.STAGE_GLOBAL] and per_cu(lds[A.STAGE_AUTO]) >= 3
    if case == "carlike_n50_candidates":
        assert lds[A.STAGE_AUTO] == lds[A.STAGE_LDS]           # the headline grid keeps everything in LDS


def test_integral_form_fixed_grid_golden(m):
    """quadratic INTEGRAL-form cost (quadratic_cost_se2.cpp:54-83) on the fixed-dt grid (weights x dt); the variable-grid case is test_integral_form_free_dt_golden."""
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_integral_n20.npz"))
    s = m.BatchSolver(m.config_unicycle_quadratic(20, integral_form=True), max_batch=g["x0"].shape[0])
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6
    assert (np.abs(r.iters - g["iters"]) <= 2).all()
    s.close()



def test_closed_loop_golden_config1(m):
    """SURVEY 8c level 3 (closed loop): 40 control cycles of BASELINE config 1 generated by the numpy oracle (plant advanced with
    u_0, fixed-grid shifted warm start); the cycles are independent given their stored inputs, so they form one GPU batch.
    First cycle = cold start (its stored guess is the reference cold start), the others start from the shifted solution."""
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_closed_loop_n20.npz"))
    B = g["x0"].shape[0]
    s = m.BatchSolver(m.config_unicycle_quadratic(20), max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], init=(g["x_init"], g["u_init"], g["dt_init"]))
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6
    assert (np.abs(r.iters - g["iters"]) <= 2).all()
    # the commanded controls of the whole run (what the robot would have executed)
    assert np.abs(r.u[:, 0] - g["u"][:, 0]).max() < 1e-7
    s.close()


def test_config3_shape_obstacle_golden(m):
    """BASELINE config 3 shape (unicycle quadratic form, n = 80, 16 polygons, <= 4 clearance rows per grid point): 24 instances of
    the numpy oracle (tests/golden/make_golden.py --config3); association on the device must pick the same rows."""
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_obstacles_n80.npz"))
    B, O, V = g["x0"].shape[0], g["vertices"].shape[1], g["vertices"].shape[2]
    s = m.BatchSolver(m.config_unicycle_quadratic(80, max_obstacles=O, max_vertices=V, max_obstacle_rows=int(g["max_rows"])), max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"]))
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6
    assert (np.abs(r.iters - g["iters"]) <= 2).all()
    s.close()


def test_long_horizon_bicycle_vs_c_oracle(m, c_oracle):
    """BASELINE config 5 shape in fp64 (kinematic bicycle, n = 120 > one wavefront of grid points, dt free): the chunked lane-parallel
    passes (two chunks of 64) must give the same iterates as the oracle -- regression test for an update-order bug in accept()
    (a rate row at a chunk boundary read its neighbour's already-updated control)."""
    from oracle import se2_nlp as R
    n, B = 120, 64
    x0, xf, up, dtp = m.workloads.bicycle_min_time_inputs(B)
    s = m.BatchSolver(m.config_bicycle_min_time(n), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    oc = c_oracle.from_nlp_config(R.config_bicycle_min_time(n))
    xo, uo, do, st, it = c_oracle.solve_batch(oc, x0, xf, up, dtp)
    both = (r.status == 0) & (st == 0)
    assert both.sum() >= 0.5 * B and abs(int((r.status == 0).sum()) - int((st == 0).sum())) <= 0.1 * B
    err = np.abs(r.x - xo).reshape(B, -1).max(1)
    assert np.median(err[both]) < 1e-8
    s.close()


@pytest.mark.parametrize("ls", ["filter", "merit"])
def test_restoration_on_the_device_follows_the_c_oracle(m, c_oracle, ls):
    """r05: the workload of tests/test_oracle_solver.py::test_restoration_for_jammed_clearance_rows_in_both_cpu_solvers (car-like minimum time, n = 30, three
    point obstacles 0.05 .. 0.5 m beside the path, d_min 0.3, reference path alone) on the device, in the LDS form and -- forced -- in the global form: the instances whose iterate
    path the restoration mode changes are known from the C oracle (its experiment switch turns the mode off); on those, and on the whole batch, the device returns the C oracle's
    statuses, its trajectories and (within a few) its iteration counts.  The workload also guards the pivot test of the root system (mpc_core.hpp::riccati_root): measured against the
    largest entry of the system instead of the pivot's own row, it ended 6 of these 64 solves with MPC_LINSOLVE in their last iterations (r05).  Under both line searches (r06): the
    restoration entry resets the filter / the l1 penalty."""
    import ctypes as C
    from oracle import se2_nlp as R
    from mpc_local_planner_amd import _abi as A
    B, n, O = 64, 30, 3
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=77, goal_range=(2.0, 4.0))
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
        lib.oracle_set_algo(C.c_int(10), C.c_double(0.0))
        off = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg, line_search=0 if ls == "merit" else 1), x0, xf, up, dtp, obstacles=obstacles, obst=ob)
    finally:
        lib.oracle_set_algo(C.c_int(10), C.c_double(1000.0))
    changed = (on[3] != off[3]) | (on[4] != off[4])
    assert changed.sum() >= 4
    for mode in (A.STAGE_AUTO, A.STAGE_GLOBAL):
        s = m.BatchSolver(m.config_carlike_min_time(n, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=O, max_vertices=1, max_obstacle_rows=4,
                                                    stage_data=mode, line_search=A.LS_MERIT if ls == "merit" else A.LS_DEFAULT), max_batch=B)
        r = s.solve(x0, xf, up, dtp, obstacles=obstacles)
        s.close()
        both = (r.status == 0) & (on[3] == 0)
        err = np.abs(r.x - on[0]).reshape(B, -1).max(1)
        print(f"[restoration on the device, {ls} line search, stage_data {mode}] converged device / oracle {int((r.status == 0).sum())} / {int((on[3] == 0).sum())} (oracle without the mode: {int((off[3] == 0).sum())}); "
              f"on the {int(changed.sum())} instances the mode changes: same status {int((r.status[changed] == on[3][changed]).sum())}, same iterations {int((r.iters[changed] == on[4][changed]).sum())}, "
              f"max |x - oracle| {err[changed & both].max():.1e}")
        assert (r.status == on[3]).mean() >= 0.97 and (r.status[changed] == on[3][changed]).sum() >= changed.sum() - 1
        # iteration counts: the end game of these solves is badly scaled (active rows: dt-dt entries of 1e11 in the value function), the last digits of the optimality error
        # differ between the sweeps and the banded LU and with them the iteration in which it passes tol -- a handful of instances end one to four iterations apart
        assert (np.abs(r.iters - on[4])[both] <= 4).mean() >= 0.95 and (r.iters == on[4])[both].mean() >= 0.8
        same = both & (r.iters == on[4])
        assert np.median(err[both]) < 1e-9 and (err[same] < 1e-5).sum() >= same.sum() - 2 and (err[both] < 1e-3).sum() >= both.sum() - 1


def test_active_clearance_rows_vs_c_oracle(m, c_oracle):
    """Obstacles INSIDE the clearance band (workload lateral=(0.15, 0.8): the rows start violated and about a quarter of the instances
    end with binding clearance constraints).  The device must converge on the same instances as the C oracle and land on the same
    solutions; a hard workload for the interior-point method itself (~80 % converge within 100 iterations in every implementation)."""
    from oracle import se2_nlp as R
    n, B, O, V, M = 80, 128, 16, 6, 4
    x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=(0.15, 0.8))
    s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
    ocfg = R.config_unicycle_quadratic(n)
    oc = c_oracle.from_nlp_config(ocfg)
    xo, uo, do, st, it = c_oracle.solve_batch(oc, x0, xf, up, dtp, obstacles=(no, nv, verts), obst=c_oracle.obst_from_nlp_config(ocfg, O, V, M))
    xn, _, _, stn, _ = c_oracle.solve_batch(oc, x0, xf, up, dtp)                      # the same instances without obstacles
    both = (r.status == 0) & (st == 0)
    # r03: clearance rows start with a slack of max(-g, 0.5): >= 95 % of these instances converge (81 % with the 1e-2 of the linear rows)
    assert (r.status == 0).mean() >= 0.95 and (st == 0).mean() >= 0.95, ((r.status == 0).mean(), (st == 0).mean())
    assert abs(int((r.status == 0).sum()) - int((st == 0).sum())) <= 0.04 * B
    err = np.abs(r.x - xo).reshape(B, -1).max(1)
    assert np.median(err[both]) < 1e-7
    moved = both & (stn == 0) & (np.abs(xo - xn).reshape(B, -1).max(1) > 1e-3)            # instances whose solution the obstacles shape
    assert moved.sum() >= 10
    assert np.median(err[moved]) < 1e-6
    _account("config 3 shape, polygons inside the clearance band, B=128", ocfg, (x0, xf, up, dtp), r, (xo, uo, do, st, it), obstacles=(no, nv, verts), max_rows=M, min_match=0.85)
    s.close()


@pytest.mark.parametrize("lateral", [(0.15, 0.8), (0.3, 1.5)])
def test_config3_full_batch_accounting(m, c_oracle, lateral):
    """BASELINE configs[2] at its full batch (B = 4096, unicycle quadratic form, n = 80, 16 polygons), both placements of bench.py's legs: every converged
    device result is within 1e-4 of the C oracle's or a KKT point of the reference-form NLP on its own; nothing unclassified (SLOW_TIER)."""
    from oracle import se2_nlp as R
    n, B, O, V, M = 80, 4096, 16, 6, 4
    x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=lateral)
    s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
    s.close()
    ocfg = R.config_unicycle_quadratic(n)
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, obstacles=(no, nv, verts), obst=c_oracle.obst_from_nlp_config(ocfg, O, V, M))
    assert (r.status == 0).mean() >= (0.95 if lateral[0] < 0.3 else 0.995)
    _account(f"config 3, lateral {lateral}, B={B}", ocfg, (x0, xf, up, dtp), r, ref, obstacles=(no, nv, verts), max_rows=M, min_match=0.9)


COLLOC_CASES = {
    "carlike_min_time_midpoint_n20": ("carlike", 20, 1), "unicycle_quadratic_midpoint_n20": ("unicycle", 20, 1), "bicycle_min_time_midpoint_n30": ("bicycle", 30, 1),
    "carlike_min_time_cn_n20": ("carlike", 20, 2), "unicycle_quadratic_cn_n20": ("unicycle", 20, 2), "bicycle_min_time_cn_n30": ("bicycle", 30, 2),
}


@pytest.mark.parametrize("name", sorted(COLLOC_CASES))
def test_midpoint_and_crank_nicolson_collocation_golden(m, name):
    """midpoint_differences (fd_collocation_se2.h:91-108) and crank_nicolson_differences (:130-147, restated literally) collocation:
    numpy-oracle fixtures in the explicit solver form (tests/golden/make_golden.py --midpoint / --cn); every fixture and every device
    result is feasible in the REFERENCE-form rows (dynamics evaluated at interpolate_angle(..) / at x_{k+1}).  Same iterate sequence."""
    from oracle import se2_nlp as R
    kind, n, method = COLLOC_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    mk = {"carlike": m.config_carlike_min_time, "unicycle": m.config_unicycle_quadratic, "bicycle": m.config_bicycle_min_time}[kind]
    s = m.BatchSolver(mk(n, collocation=method), max_batch=g["x0"].shape[0])
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6 and np.abs(r.dt - g["dt"]).max() < 1e-8
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all()
    ocfg = {"carlike": R.config_carlike_min_time, "unicycle": R.config_unicycle_quadratic, "bicycle": R.config_bicycle_min_time}[kind](n)
    ocfg.collocation = method
    for i in range(g["x0"].shape[0]):
        nlp = R.ReferenceNlp(ocfg, R.CycleInputs(x0=g["x0"][i], xf=g["xf"][i], u_prev=g["u_prev"][i], dt_prev=float(g["dt_prev"][i])))
        assert np.abs(nlp.equalities(nlp.pack(R.Trajectory(r.x[i], r.u[i, :-1], float(r.dt[i]))))).max() < 1e-6
    s.close()


@pytest.mark.parametrize("method", [1, 2])
def test_config2_midpoint_and_crank_nicolson_batch_vs_c_oracle(m, c_oracle, method):
    """BASELINE.json config 2 (car-like min-time, n=50, B=256 of the 8d input distribution) with the other two collocation rules of
    grid/collocation_method (src/controller.cpp:298-312) against oracle/mpc_oracle.c's stage_map restatement."""
    import copy
    from oracle import se2_nlp as R
    B = 256
    _, ocfg = _cases(m)["carlike_min_time_n50"]
    ocfg = copy.deepcopy(ocfg)
    ocfg.collocation = method
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(m.config_carlike_min_time(50, collocation=method), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp)
    both = (r.status == 0) & (st == 0)
    assert both.mean() > 0.8
    assert (r.status == st).mean() > 0.93
    err = np.maximum(np.abs(r.x - xo).reshape(B, -1).max(1), np.abs(r.u - uo).reshape(B, -1).max(1))
    err = np.maximum(err, np.abs(r.dt - do))
    assert np.median(err[both]) < 1e-8
    _account(f"config 2, collocation {method}, B=256", ocfg, (x0, xf, up, dtp), r, (xo, uo, do, st, it))
    for i in np.nonzero(r.status == 0)[0][:32]:
        assert _feasibility(R, ocfg, x0, xf, up, dtp, r, i) < 1e-6
    s.close()


def test_terminal_ball_golden(m):
    """a18 TerminalBallSE2 (final_state_conditions_se2.cpp:54-64, configured at src/controller.cpp:676-703): l2-ball row on the free final
    state.  Fixture (tests/golden/make_golden.py --ball): effort-dominated quadratic form whose unconstrained solution stops short of
    the goal, so the row is ACTIVE in every instance.  Plus: a ball that is far from binding (the reference's example radius 5) leaves
    the config-1 goldens untouched."""
    from oracle import se2_nlp as R
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_ball_n20.npz"))
    B = g["x0"].shape[0]
    cfg = m.config_unicycle_quadratic(20, Q=tuple(g["Q"]), R=tuple(g["R"]), Qf=None, terminal_ball_S=tuple(g["S"]), terminal_ball_gamma=float(g["gamma"]))
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all()
    # (the states of these effort-dominated solutions have flat directions: dense numpy and the Riccati sweeps stop 5e-6 apart at a KKT error of 1e-9)
    assert np.abs(r.x - g["x"]).max() < 1e-5 and np.abs(r.u - g["u"]).max() < 1e-5
    assert (np.abs(r.iters - g["iters"]) <= 2).all()
    xd = r.x[:, -1] - g["xf"]
    xd[:, 2] = R.normalize_theta(xd[:, 2])
    val = (xd * xd * g["S"]).sum(1)
    assert np.abs(val - float(g["gamma"])).max() < 1e-6                      # on the ball's boundary ...
    xdf = g["x_free"][:, -1] - g["xf"]
    assert ((xdf[:, :2] ** 2).sum(1) > 2 * float(g["gamma"])).all()         # ... which the solution without the row is far outside of
    s.close()
    g1 = np.load(os.path.join(GOLD, "unicycle_quadratic_n20.npz"))
    s = m.BatchSolver(m.config_unicycle_quadratic(20, terminal_ball_S=(1.0, 1.0, 1.0), terminal_ball_gamma=5.0), max_batch=g1["x0"].shape[0])
    r = s.solve(g1["x0"], g1["xf"], g1["u_prev"], g1["dt_prev"])
    # (an inactive row is still a complementarity pair: it enters the average the adaptive barrier rule follows, and the solve stops 5e-6 away along the flat directions)
    assert (r.status == 0).all() and np.abs(r.x - g1["x"]).max() < 1e-5 and np.abs(r.u - g1["u"]).max() < 1e-4
    s.close()


@pytest.mark.parametrize("name", ["carlike_via_points_n30", "carlike_via_points_ordered_n30"])
def test_via_points_objective_golden(m, name):
    """minimum_time_via_points (src/optimal_control/min_time_via_points_cost.cpp:39-145): association of every via-point
    with its closest grid point of the starting trajectory (plain and ordered mode, incl. the 'behind the start' branch), quadratic
    attraction on the position, the orientation term linear as the reference codes it.  Fixtures: tests/golden/make_golden.py --via."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, VP = g["x0"].shape[0], g["via"].shape[1]
    cfg = m.config_carlike_min_time(30, objective=m.OBJ_MIN_TIME_VIA_POINTS, vp_position_weight=float(g["wp"]), vp_orientation_weight=float(g["wo"]),
                                    via_points_ordered=bool(g["ordered"]), max_via_points=VP)
    s = m.BatchSolver(cfg, max_batch=B)
    s.set_via_points(g["n_via"], g["via"])
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all()
    assert np.abs(r.x - g["x"]).max() < 1e-6 and np.abs(r.u - g["u"]).max() < 1e-6 and np.abs(r.dt - g["dt"]).max() < 1e-7
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all()
    # cleared via-points: plain minimum time (shorter transition time than with the detours)
    s.set_via_points(None)
    p = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    ok = p.status == 0
    # (local solves: one instance may end in a basin whose plain minimum-time optimum is a hair slower than the detour's -- seen with 1e-4 s -- so all but one)
    assert ok.sum() >= B - 1 and (p.dt[ok] < r.dt[ok] + 1e-9).sum() >= ok.sum() - 1 and (p.dt[ok] < r.dt[ok] - 1e-4).any()
    s.close()


def test_costmap_to_point_obstacles_bit_exact(m):
    """MpcLocalPlannerROS::updateObstacleContainerWithCostmap (src/mpc_local_planner_ros.cpp:474-499) on the device vs
    oracle/costmap.py: same obstacles, same ORDER, bit-identical coordinates; capacity overflow is reported; edge cases: empty map,
    lethal cells only in the last row / column (never visited), a map narrower than one slab and one wider than 256 columns."""
    from oracle import costmap as OC
    rng = np.random.default_rng(20260927)
    for (sx, sy, B, dens, O) in ((60, 45, 5, 0.02, 256), (300, 70, 3, 0.004, 256), (40, 40, 2, 0.2, 64), (1, 1, 1, 1.0, 8), (2, 5, 2, 1.0, 8)):
        pr = (1.0 - dens) * np.array([0.5, 0.2, 0.1, 0.1, 0.0, 0.1]); pr[4] = dens
        cost = rng.choice(np.array([0, 1, 128, 253, 254, 255], np.uint8), size=(B, sy, sx), p=pr).astype(np.uint8)
        if sx > 10:
            cost[0] = 0
            cost[0, -1, :] = 254; cost[0, :, -1] = 254            # instance 0: lethal only where the reference never looks
        res = 0.05
        origin = rng.uniform(-5, 5, (B, 2))
        pose = np.concatenate([origin + rng.uniform(0.2, 0.8, (B, 2)) * np.array([sx, sy]) * res, rng.uniform(-np.pi, np.pi, (B, 1))], 1)
        s = m.BatchSolver(m.config_unicycle_quadratic(20, max_obstacles=O, max_vertices=3), max_batch=B)
        no, nv, vt, dr = s.costmap_to_obstacles(cost, res, origin, pose, behind_robot_dist=0.6)
        for b in range(B):
            ref = OC.costmap_to_obstacles(cost[b], res, origin[b], pose[b], 0.6)
            k = min(len(ref), O)
            assert no[b] == k and dr[b] == len(ref) - k
            assert (nv[b, :k] == 1).all() and (nv[b, k:] == 0).all()
            np.testing.assert_array_equal(vt[b, :k, 0, :], ref[:k])
        if sx > 10:
            assert no[0] == 0
        s.close()


def test_costmap_obstacles_feed_the_solve(m):
    """costmap cells -> point obstacles -> clearance rows, all through the C ABI: the solved trajectory keeps min_obstacle_dist to every
    lethal cell centre it was given."""
    from oracle import costmap as OC
    sx = sy = 80
    res = 0.05
    cost = np.zeros((1, sy, sx), np.uint8)
    cost[0, 44:48, 40:42] = 254                                 # a block whose nearest cell centre is ~0.17 m beside the straight line start -> goal
    origin = np.array([[-0.5, -2.0]])
    x0 = np.array([[0.0, 0.0, 0.0]]); xf = np.array([[3.0, 0.1, 0.0]])
    s = m.BatchSolver(m.config_unicycle_quadratic(40, max_obstacles=64, max_vertices=1), max_batch=1)
    no, nv, vt, dr = s.costmap_to_obstacles(cost, res, origin, x0, behind_robot_dist=1.5)
    assert no[0] == 8 and dr[0] == 0
    r = s.solve(x0, xf, np.zeros((1, 2)), np.array([0.2]), obstacles=(no, nv, vt))
    assert r.status[0] == 0
    pts = OC.costmap_to_obstacles(cost[0], res, origin[0], x0[0])
    d = np.sqrt(((r.x[0, 1:-1, None, :2] - pts[None]) ** 2).sum(-1))
    assert d.min() > 0.2 - 1e-6 and d.min() < 0.2 + 1e-3        # min_obstacle_dist of the config; the block is in the way, so the row binds
    s.close()


@pytest.mark.parametrize("ordered", [False, True])
def test_via_points_batch_vs_c_oracle(m, c_oracle, ordered):
    """config-2 family (car-like, n=50) with 3 random via-points per instance, B=192: association (incl. skipped / clamped points) and
    objective terms against oracle/mpc_oracle.c at batch scale."""
    import copy
    B, VP, n = 192, 4, 50
    _, ocfg = _cases(m)["carlike_min_time_n50"]
    ocfg = copy.deepcopy(ocfg)
    ocfg.objective, ocfg.vp_position_weight, ocfg.vp_orientation_weight, ocfg.via_points_ordered = 2, 10.5, 0.0, ordered
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=777)
    rng = np.random.default_rng(778)
    t = rng.uniform(-0.1, 1.1, (B, 3, 1))                         # a few lie before the start / behind the goal
    if ordered:
        t = np.sort(t, axis=1)
    d = (xf[:, None, :2] - x0[:, None, :2])
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    via = np.zeros((B, VP, 3))
    via[:, :3, :2] = x0[:, None, :2] + t * d + rng.uniform(-0.4, 0.4, (B, 3, 1)) * nrm
    nvia = rng.integers(0, 4, B).astype(np.int32)                # 0..3 via-points
    s = m.BatchSolver(m.config_carlike_min_time(n, objective=m.OBJ_MIN_TIME_VIA_POINTS, vp_position_weight=10.5, via_points_ordered=ordered,
                                                max_via_points=VP), max_batch=B)
    s.set_via_points(nvia, via)
    r = s.solve(x0, xf, up, dtp)
    xo, uo, do, st, it = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), x0, xf, up, dtp, via=(nvia, via))
    both = (r.status == 0) & (st == 0)
    assert both.mean() > 0.75 and (r.status == st).mean() > 0.9
    err = np.maximum(np.abs(r.x - xo).reshape(B, -1).max(1), np.abs(r.u - uo).reshape(B, -1).max(1))
    # same algorithm and association; the via-point problems are flat around the solution, so a flipped line-search tie shows up as ~1e-4
    assert np.median(err[both]) < 1e-7 and (err[both] < 1e-3).mean() > 0.9
    assert (np.abs(r.iters - it)[both] <= 2).mean() > 0.8
    s.close()


def test_line_footprint_golden(m):
    """a21 with teb's LineRobotFootprint (the car-like example's footprint, cfg/carlike/mpc_local_planner_params.yaml:19-23) against point
    obstacles: the clearance rows depend on the heading (position-heading coupling in gradient and Hessian, A-form slots A02/A12).
    Fixture: tests/golden/make_golden.py --line (numpy oracle, analytic row derivatives checked against differences); half of the
    instances end with a binding row."""
    from oracle import se2_nlp as R
    g = np.load(os.path.join(GOLD, "carlike_line_footprint_n30.npz"))
    B, O = g["pts"].shape[0], g["pts"].shape[1]
    cfg = m.config_carlike_min_time(30, footprint_kind=2, footprint_params=tuple(g["line"]), min_obstacle_dist=0.27, force_inclusion_dist=0.5,
                                    cutoff_dist=2.5, max_obstacles=O, max_vertices=1, max_obstacle_rows=int(g["max_rows"]))
    s = m.BatchSolver(cfg, max_batch=B)
    no = np.full(B, O, np.int32); nv = np.ones((B, O), np.int32); vt = g["pts"].reshape(B, O, 1, 2)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(no, nv, vt))
    assert (r.status == 0).all()
    err = np.maximum(np.abs(r.x - g["x"]).reshape(B, -1).max(1), np.abs(r.u - g["u"]).reshape(B, -1).max(1))
    same = r.iters == g["iters"]
    assert same.sum() >= B - 1 and (err[same] < 1e-5).all() and np.median(err) < 1e-8      # flat problems: the same iteration count can stop a few 1e-6 apart (north-star bound: 1e-4)
    assert np.abs(r.dt - g["dt"]).max() < 1e-6
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all()
    for i in range(B):                       # clearance of the FOOTPRINT (not of the reference point) in the reference's own distance function
        obs = [R.Obstacle(R.OBST_POINT, g["pts"][i, o:o + 1]) for o in range(O)]
        dmin = min(R.footprint_distance(R.FOOTPRINT_LINE, tuple(g["line"]), r.x[i, k], ob) for k in range(1, 29) for ob in obs)
        assert dmin > 0.27 - 1e-6
    s.close()


def test_two_circles_footprint_golden(m):
    """a21 with teb's TwoCirclesRobotFootprint against polygon obstacles (fixture: tests/golden/make_golden.py --two; three of the six
    instances end with a binding row); the footprint clearance is re-checked with the reference-form distance function."""
    from oracle import se2_nlp as R
    g = np.load(os.path.join(GOLD, "unicycle_two_circles_obstacles_n30.npz"))
    B, O, V = g["vertices"].shape[0], g["vertices"].shape[1], g["vertices"].shape[2]
    cfg = m.config_unicycle_quadratic(30, footprint_kind=3, footprint_params=tuple(g["two"]), max_obstacles=O, max_vertices=V, max_obstacle_rows=int(g["max_rows"]))
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"]))
    assert (r.status == 0).all()
    err = np.maximum(np.abs(r.x - g["x"]).reshape(B, -1).max(1), np.abs(r.u - g["u"]).reshape(B, -1).max(1))
    same = r.iters == g["iters"]
    assert same.sum() >= B - 2 and (err[same] < 1e-6).all() and (err < 1e-4).all()
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all()
    for i in range(B):
        obs = [R.Obstacle(R.OBST_POLYGON, g["vertices"][i, o, :g["n_vertices"][i, o]]) for o in range(g["n_obstacles"][i])]
        dmin = min(R.footprint_distance(R.FOOTPRINT_TWO_CIRCLES, tuple(g["two"]), r.x[i, k], ob) for k in range(1, 29) for ob in obs)
        assert dmin > 0.2 - 1e-6
    s.close()


def test_integral_form_free_dt_golden(m):
    """a17 on the variable grid: dt * sum(xd'Q xd + u'R u) with dt a decision variable (quadratic_cost_se2.cpp:54-83, left sum
    finite_differences_grid_se2.cpp:61-75): state-dt and control-dt coupling in the Hessian (A-form slots A05 A15 A25 A56 A57), the stage
    costs in the dt gradient.  Fixture: tests/golden/make_golden.py --integral-free."""
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_integral_free_dt_n20.npz"))
    B = g["x0"].shape[0]
    cfg = m.config_unicycle_quadratic(20, dt_free=True, dt_lb=0.01, dt_ub=2.0, xf_fixed=(True, True, True), Qf=None, integral_form=True, R=(1.0, 0.5))
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (r.status == 0).all()
    err = np.maximum(np.abs(r.x - g["x"]).reshape(B, -1).max(1), np.abs(r.u - g["u"]).reshape(B, -1).max(1))
    same = r.iters == g["iters"]
    assert same.sum() >= B - 1 and (err[same] < 1e-6).all() and (err < 1e-4).all()
    assert np.abs(r.dt - g["dt"]).max() < 1e-6
    s.close()


def test_dynamic_obstacles_golden(m):
    """a22 computeNonIntegralStateDtTerm (stage_inequality_se2.cpp:177-189) + the 'always kept' association of dynamic obstacles (:99-106):
    a circular obstacle crossing the path with constant velocity (row at grid point k against the position predicted for t = k dt -> dt
    enters gradient and Hessian of the row) and a static point obstacle; car-like min-time, dt free.  Fixture: make_golden.py --dynamic
    (all instances end with the moving obstacle's row binding)."""
    from oracle import se2_nlp as R
    g = np.load(os.path.join(GOLD, "carlike_dynamic_obstacles_n30.npz"))
    B, O = g["x0"].shape[0], g["vertices"].shape[1]
    cfg = m.config_carlike_min_time(30, enable_dynamic_obstacles=True, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5,
                                    max_obstacles=O, max_vertices=1, max_obstacle_rows=int(g["max_rows"]))
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"], g["radius"], g["velocity"]))
    assert (r.status == 0).all()
    err = np.maximum(np.abs(r.x - g["x"]).reshape(B, -1).max(1), np.abs(r.u - g["u"]).reshape(B, -1).max(1))
    same = r.iters == g["iters"]
    assert same.sum() >= B - 1 and (err[same] < 1e-6).all() and (err < 1e-4).all()
    assert np.abs(r.dt - g["dt"]).max() < 1e-5
    for i in range(B):          # clearance to the MOVING obstacle along the solved trajectory, reference-form distance at t = k dt
        ob = R.Obstacle(R.OBST_CIRCLE, g["vertices"][i, 0], radius=float(g["radius"][i, 0]), velocity=g["velocity"][i, 0])
        dmin = min(R.footprint_distance(R.FOOTPRINT_POINT, (), r.x[i, k], ob, k * r.dt[i]) for k in range(1, 29))
        assert abs(dmin - 0.3) < 1e-5
    # the same obstacles frozen (velocity ignored): a different, faster trajectory
    s2 = m.BatchSolver(m.config_carlike_min_time(30, min_obstacle_dist=0.3, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=O, max_vertices=1,
                                                 max_obstacle_rows=int(g["max_rows"])), max_batch=B)
    q = s2.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(g["n_obstacles"], g["n_vertices"], g["vertices"], g["radius"]))
    assert (np.abs(q.x - r.x).reshape(B, -1).max(1) > 1e-3).all()
    s.close(); s2.close()


def test_polygon_footprint_golden(m):
    """a21 with teb's PolygonRobotFootprint (the vertex list of cfg/carlike/mpc_local_planner_params.yaml:28) against point obstacles: the
    obstacle in the robot frame against the closed edge loop (first closest edge, no inside test).  Fixture: make_golden.py --polygon
    (four of five instances end with a binding row)."""
    from oracle import se2_nlp as R
    g = np.load(os.path.join(GOLD, "carlike_polygon_footprint_n30.npz"))
    B, O = g["pts"].shape[0], g["pts"].shape[1]
    cfg = m.config_carlike_min_time(30, footprint_kind=4, footprint_vertices=tuple(g["poly"]), min_obstacle_dist=0.15, force_inclusion_dist=0.5,
                                    cutoff_dist=2.5, max_obstacles=O, max_vertices=1, max_obstacle_rows=int(g["max_rows"]))
    s = m.BatchSolver(cfg, max_batch=B)
    no = np.full(B, O, np.int32); nv = np.ones((B, O), np.int32); vt = g["pts"].reshape(B, O, 1, 2)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"], obstacles=(no, nv, vt))
    assert (r.status == 0).all()
    err = np.maximum(np.abs(r.x - g["x"]).reshape(B, -1).max(1), np.abs(r.u - g["u"]).reshape(B, -1).max(1))
    same = r.iters == g["iters"]
    assert same.sum() >= B - 2 and (err[same] < 1e-6).all() and (err < 1e-4).all()
    assert (np.abs(r.iters - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all()
    for i in range(B):
        obs = [R.Obstacle(R.OBST_POINT, g["pts"][i, o:o + 1]) for o in range(O)]
        dmin = min(R.footprint_distance(R.FOOTPRINT_POLYGON, tuple(g["poly"]), r.x[i, k], ob) for k in range(1, 29) for ob in obs)
        assert dmin > 0.15 - 1e-6
    s.close()


def test_candidate_initial_trajectories_best_of(m):
    """north star: "batches of independent planner instances (and candidate initial trajectories)".  The reference's two initialisations
    (2-pose-plan cold start; initializeSequences without xinit) as candidates of every config-2 instance, solved as ONE batch of 2B
    members and reduced on the host (mpc_local_planner_amd/candidates.py; the CPU twin of this test runs the same code on the C oracle,
    tests/test_candidates.py): more instances converge, none ends worse than its cold-start candidate."""
    from mpc_local_planner_amd import candidates as K
    B = 256
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=2 * B)
    best, win, allr = K.solve_best_of(s, x0, xf, up, dtp, guesses=("cold", "travel"))
    cold_ok = allr.status[:B] == 0
    assert (best.status == 0).mean() >= 0.97 and (best.status == 0).sum() >= min(B, cold_ok.sum() + 3)      # the C solver's cold start: 251 of these 256
    assert (best.status[cold_ok] == 0).all() and (best.dt[cold_ok] <= allr.dt[:B][cold_ok] + 1e-12).all()
    single = s.solve(x0, xf, up, dtp)                                   # the device-side cold start is the same guess
    same = cold_ok & (single.status == 0)
    err = np.abs(allr.x[:B, :, :2] - single.x[:, :, :2]).reshape(B, -1).max(1)[same]
    assert np.median(err) < 1e-7
    s.close()


@pytest.mark.parametrize("which", [2, 3, "3b", "3c"])
def test_device_vs_independent_sqp_from_the_cold_start(m, which):
    """SURVEY 8c level 2 on the device: scipy SLSQP on the reference-form NLP from the reference cold start (fixtures of
    tests/golden/make_cold_start_scipy.py, 32 instances each of config 2 and config 3; r04: 24 each of config 3 on the two placements where the clearance rows bind, 3b / 3c)
    against the HIP solve."""
    from test_oracle_solver import _vs_independent_sqp

    def solve(ocfg, inputs, obst):
        if obst is None:
            s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=32)
            r = s.solve(*inputs)
        else:
            s = m.BatchSolver(m.config_unicycle_quadratic(80, max_obstacles=16, max_vertices=6, max_obstacle_rows=4), max_batch=32)
            r = s.solve(*inputs, obstacles=obst)
        s.close()
        return r.x, r.u, r.dt, r.status, r.iters
    same, err, conv = _vs_independent_sqp(f"device vs SLSQP, config {which}", which, solve)
    if which == 3:
        assert conv.all() and same.all()
    elif which == "3b":
        assert conv.sum() >= len(conv) - 2 and same.sum() >= 0.8 * conv.sum()
    elif which == "3c":
        assert conv.sum() >= 0.3 * len(conv) and same.sum() >= 0.5 * conv.sum()
    else:
        assert conv.sum() >= 28 and same.sum() >= 14          # r04: 29 converged, 16 at SLSQP's point (r03: 30 / 14)


@pytest.mark.gpu
def test_terminal_cost_with_minimum_time_objective_vs_c_oracle(m, c_oracle):
    """planning/terminal_cost quadratic with planning/objective minimum_time and a free final state (src/controller.cpp:641-672,
    finite_differences_grid_se2.cpp:128-133): device against the C oracle, accounting over B = 128."""
    import dataclasses
    from oracle import se2_nlp as R
    B, n = 128, 30
    ocfg = dataclasses.replace(R.config_carlike_min_time(n), Qf=np.array([2.0, 2.0, 2.0]), xf_fixed=(False, False, False), dt_lb=0.05)
    cfg = m.config_carlike_min_time(n, Qf=(2.0, 2.0, 2.0), xf_fixed=(False, False, False), dt_lb=0.05)
    inputs = m.workloads.carlike_min_time_inputs(B, seed=31, goal_range=(1.0, 4.0))
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(*inputs)
    out = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *inputs)
    match, other = _account("min-time + terminal cost, free xf", ocfg, inputs, r, out)
    assert (r.status == 0).mean() > 0.9 and match.sum() > 0.85 * B
    assert np.abs(r.x[:, -1] - inputs[1]).max(1)[r.status == 0].max() > 0.05      # the terminal cost is live: the final state is not the goal
    s.close()


def test_device_answers_are_local_minima(m):
    """r04: the kernel accepts a factorisation on its inertia (signs of the Riccati sweep's control pivots + the (dt, nu) root system, mpc_core.hpp::riccati_root) -- Ipopt's test.
    Checked on the device's own answers with nothing of any solver: the reduced Hessian of the Lagrangian of the reference-form NLP on the tangent space of the active rows
    (oracle/kkt_check.py::second_order, differences of the NLP's functions) has no negative eigenvalue at any of 24 converged answers of the headline workload -- reference
    path and hedges alike.  (With the inertia-free curvature test of r01-r03 about one converged answer in five was a saddle point:
    tests/test_oracle_solver.py::test_converged_answers_are_local_minima_and_the_curvature_test_s_were_not.)"""
    from oracle import kkt_check as KC
    from mpc_local_planner_amd import _abi as A
    B, n = 96, 50
    _, ocfg = _cases(m)["carlike_min_time_n50"]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    s = m.BatchSolver(m.config_carlike_min_time(n, candidates=(A.CAND_REFERENCE, A.CAND_HERMITE_FF, A.CAND_HERMITE_FF, A.CAND_HERMITE_FR), candidate_max_iter=(100, 45, 40, 35),
                                                candidate_param=(0.0, 2.0, 3.0, 1.5)), max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    win, _ = s.last_candidates(B)
    s.close()
    ok = np.nonzero(r.status == 0)[0]
    hedged = [i for i in ok if win[i] > 0][:6]
    pick = sorted(set(hedged) | set(ok[:24 - len(hedged)].tolist()))
    so = KC.second_order_many(ocfg, x0, xf, up, dtp, r.x, r.u, r.dt, pick)
    eig = np.array([so[i]["min_eig_s"] for i in pick])
    print(f"[second-order check of device answers] {len(pick)} converged answers ({len(hedged)} supplied by a hedge): tangent-space dimensions {sorted(so[i]['dim_s'] for i in pick)}, "
          f"smallest eigenvalue of the reduced Hessian {np.min(eig):.2e} (vertices of the active set count as +inf), weakly active rows {max(so[i]['n_weak'] for i in pick)}")
    assert len(pick) >= 20 and (eig > -1e-6).all()


def test_time_limit_per_solve(m):
    """solver/ipopt/max_cpu_time (src/controller.cpp:395-397 -> SolverIpopt::setMaxCpuTime) = mpc_config.max_time_us, a budget per solve on the device's clock: a generous budget
    changes nothing, bit for bit; a budget of 200 us ends every solve that is not done by then with MPC_TIME_LIMIT after a handful of iterations (one iteration ~45 us), the rest
    converged as before; the reference's wrapper counts MPC_TIME_LIMIT as a failed solve like any other status but 0."""
    from mpc_local_planner_amd import _abi as A
    B, n = 256, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    out = {}
    for tag, t in (("none", 0.0), ("1 s", 1.0), ("200 us", 200e-6)):
        s = m.BatchSolver(m.config_carlike_min_time(n, max_cpu_time=t), max_batch=B)
        out[tag] = s.solve(x0, xf, up, dtp)
        s.close()
    a, b, c = out["none"], out["1 s"], out["200 us"]
    assert np.array_equal(a.status, b.status) and np.array_equal(a.iters, b.iters) and np.array_equal(a.x, b.x) and np.array_equal(a.dt, b.dt)
    lim = c.status == 5
    print(f"[time limit 200 us] {int(lim.sum())} of {B} solves ran out of time after {c.iters[lim].min()} .. {c.iters[lim].max()} iterations; {(c.status == 0).sum()} converged inside the budget")
    assert A.STATUS_NAMES[5] == "time_limit" and lim.sum() >= 0.9 * B and c.iters[lim].max() <= 12 and c.iters[lim].min() >= 1
    assert set(np.unique(c.status)) <= {0, 5}
    done = c.status == 0
    assert np.array_equal(c.x[done], a.x[done]) and (a.status[done] == 0).all()
