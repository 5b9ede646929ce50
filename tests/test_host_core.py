"""CPU test of the DEVICE solver core: mpc_core.hpp (the code the HIP kernel instantiates) is compiled
for the host by a tests-only harness (tests/host_harness/host_solver.cpp) and compared with the oracle.
This is a developer check of the kernel's arithmetic without a GPU -- the harness is not part of the
package and is never loaded by it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import se2_nlp as R
from mpc_local_planner_amd import _abi as A
from mpc_local_planner_amd import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "host_solver.cpp")
OUT = os.path.join(HERE, "host_harness", "_build", "libmpc_hostdbg.so")
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def host():
    csrc = os.path.join(HERE, "..", "mpc_local_planner_amd", "csrc")
    deps = [SRC, os.path.join(HERE, "host_harness", "ipm_serial.hpp"), os.path.join(csrc, "mpc_core.hpp"), os.path.join(csrc, "mpc_problem.hpp"), os.path.join(HERE, "..", "include", "mpc_hip.h")]
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", OUT], check=True)
    return C.CDLL(OUT)


def host_solve(lib, cfg, x0, xf, up, dtp, init=None):
    B, n = x0.shape[0], cfg.n
    xo = np.zeros((B, n, 3)); uo = np.zeros((B, n, 2)); do = np.zeros(B)
    st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros(B)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    xi = ui = di = None
    if init is not None:
        xi, ui, di = (np.ascontiguousarray(a, float) for a in init)
    lib.hostdbg_solve(C.byref(cfg), C.c_int(B), p(x0), p(xf), p(up), p(dtp), p(xi), p(ui), p(di), p(xo), p(uo), p(do), p(st), p(it), p(kkt))
    return xo, uo, do, st, it, kkt


CASES = {
    "carlike_min_time_n50": lambda **k: A.config_carlike_min_time(50, **k),
    "carlike_min_time_n20": lambda **k: A.config_carlike_min_time(20, **k),
    "unicycle_quadratic_n20": lambda **k: A.config_unicycle_quadratic(20, **k),
    "bicycle_min_time_n30": lambda **k: A.config_bicycle_min_time(30, **k),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_core_on_host_reproduces_golden(host, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = CASES[name]()
    xo, uo, do, st, it, kkt = host_solve(host, cfg, g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6
    assert np.abs(uo - g["u"]).max() < 1e-6
    assert np.abs(do - g["dt"]).max() < 1e-8
    # the Riccati sweep follows the dense solve's iterate sequence (long runs may differ by a few round-off-triggered steps)
    assert (np.abs(it - g["iters"]) <= np.maximum(2, 0.1 * g["iters"])).all() and np.median(np.abs(it - g["iters"])) == 0


def test_device_core_matches_c_oracle_on_a_batch(host, c_oracle):
    n = 30
    cfg = A.config_carlike_min_time(n)
    x0, xf, up, dtp = W.carlike_min_time_inputs(48, seed=11, goal_range=(1.0, 4.0))
    a = host_solve(host, cfg, x0, xf, up, dtp)
    oc = c_oracle.from_nlp_config(R.config_carlike_min_time(n))
    b = c_oracle.solve_batch(oc, x0, xf, up, dtp)
    both = (a[3] == 0) & (b[3] == 0)
    assert both.sum() >= 40
    err = np.maximum(np.abs(a[0] - b[0]).reshape(48, -1).max(1), np.abs(a[1] - b[1]).reshape(48, -1).max(1))
    # same algorithm, different linear algebra: identical up to round-off unless a line-search decision flips
    assert (err[both] < 1e-6).mean() >= 0.9
    assert np.median(err[both]) < 1e-9


@pytest.mark.parametrize("name,n,B", [("carlike_min_time_n20", 30, 48), ("unicycle_quadratic_n20", 20, 24), ("bicycle_min_time_n30", 30, 24)])
def test_device_core_filter_line_search_follows_the_c_oracle(host, c_oracle, name, n, B):
    """mpc_config.line_search = MPC_LS_FILTER (Ipopt's filter line search, Waechter & Biegler 2006 Algorithm A without second-order correction): the kernel's statements
    (tests/host_harness/ipm_serial.hpp carries the ones of mpc_wave_solve.inc) against the C oracle's (oracle_config.line_search = 1): same statuses, the same iteration
    counts on nearly every instance, the same trajectories; and the filter is a DIFFERENT iteration from the l1 merit on some of these instances."""
    mk = {"carlike_min_time_n20": (A.config_carlike_min_time, R.config_carlike_min_time, lambda: W.carlike_min_time_inputs(B, seed=11, goal_range=(1.0, 4.0))),
          "unicycle_quadratic_n20": (A.config_unicycle_quadratic, R.config_unicycle_quadratic, lambda: W.unicycle_quadratic_inputs(B, seed=12)),
          "bicycle_min_time_n30": (A.config_bicycle_min_time, R.config_bicycle_min_time, lambda: W.bicycle_min_time_inputs(B, seed=13, goal_range=(1.0, 5.0)))}[name]
    x0, xf, up, dtp = mk[2]()
    a = host_solve(host, mk[0](n, line_search=A.LS_FILTER), x0, xf, up, dtp)
    am = host_solve(host, mk[0](n, line_search=A.LS_MERIT), x0, xf, up, dtp)
    b = c_oracle.solve_batch(c_oracle.from_nlp_config(mk[1](n), line_search=1), x0, xf, up, dtp)
    both = (a[3] == 0) & (b[3] == 0)
    assert (a[3] == b[3]).mean() >= 0.95 and both.sum() >= 0.8 * B
    err = np.maximum(np.abs(a[0] - b[0]).reshape(B, -1).max(1), np.abs(a[1] - b[1]).reshape(B, -1).max(1))
    assert (a[4] == b[4])[both].mean() >= 0.9
    assert (err[both] < 1e-6).mean() >= 0.9 and np.median(err[both]) < 1e-9
    if name != "unicycle_quadratic_n20":
        assert (a[4] != am[4]).any()
    # ... and the l1 merit (MPC_LS_MERIT) against the C oracle's (oracle_config.line_search = 0)
    bm = c_oracle.solve_batch(c_oracle.from_nlp_config(mk[1](n), line_search=0), x0, xf, up, dtp)
    both = (am[3] == 0) & (bm[3] == 0)
    errm = np.abs(am[0] - bm[0]).reshape(B, -1).max(1)
    assert (am[3] == bm[3]).mean() >= 0.95 and (am[4] == bm[4])[both].mean() >= 0.9 and np.median(errm[both]) < 1e-9


def test_device_core_fp32_is_close_to_fp64(host):
    n = 20
    x0, xf, up, dtp = W.carlike_min_time_inputs(16, seed=12, goal_range=(1.0, 2.5))
    a = host_solve(host, A.config_carlike_min_time(n), x0, xf, up, dtp)
    b = host_solve(host, A.config_carlike_min_time(n, precision=A.FP32, tol=1e-4), x0, xf, up, dtp)
    both = (a[3] == 0) & (b[3] == 0)
    assert both.sum() >= 8
    err = np.abs(a[0] - b[0]).reshape(16, -1).max(1)
    # BASELINE.md's fp32 goal is 1e-3; the pure-fp32 path currently reaches ~3e-3 (open items)
    assert np.median(err[both]) < 1e-2


def test_device_core_edge_cases(host):
    cfg = A.config_carlike_min_time(3)          # smallest grid: 2 intervals
    x0 = np.array([[0.0, 0.0, 0.0]]); xf = np.array([[0.3, 0.0, 0.0]])
    xo, uo, do, st, it, _ = host_solve(host, cfg, x0, xf, np.zeros((1, 2)), np.zeros(1))   # dt_prev = 0 -> no stage-0 rate rows
    assert st[0] == 0
    np.testing.assert_array_equal(xo[0, 0], x0[0]); np.testing.assert_array_equal(xo[0, -1], xf[0])
    assert uo[0, 0, 0] <= 0.4 + 1e-9 and do[0] > 0
    np.testing.assert_array_equal(uo[0, -1], uo[0, -2])      # duplicated last control (getStateAndControlTimeSeries)
    # heading crossing +-pi
    cfg = A.config_carlike_min_time(20)
    x0 = np.array([[0.0, 0.0, 3.0]]); xf = np.array([[-1.5, -0.2, -3.0]])
    xo, uo, do, st, it, _ = host_solve(host, cfg, x0, xf, np.zeros((1, 2)), np.full(1, 0.2))
    assert st[0] == 0
    assert np.all(xo[0, :, 2] >= -np.pi) and np.all(xo[0, :, 2] < np.pi)
    assert np.abs(xo[0, :, 2]).min() > 2.0      # went the short way round through +-pi


def test_device_trig_kernels_match_libm(host):
    """mpc_core.hpp replaces libm's sincos/tan by a pi/2 Cody-Waite reduction + minimax kernels (the general routines
    dominated the line-search trials on the GPU): <= 2 ulp for wrapped angles, a few ulp for tan inside the steering box."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-3.2, 3.2, 200000), rng.uniform(-100, 100, 50000), [0.0, -np.pi, np.pi, np.pi / 2, -np.pi / 2, 1e-300, 1e6, -1e7]])
    s = np.empty_like(x); c = np.empty_like(x); t = np.empty_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    host.hostdbg_trig(C.c_int(x.size), p(x), p(s), p(c), p(t))
    assert np.abs(s - np.sin(x)).max() < 2.5e-16 and np.abs(c - np.cos(x)).max() < 2.5e-16
    w = np.abs(x) < 1.45
    assert (np.abs(t[w] - np.tan(x[w])) <= 4 * np.spacing(np.abs(np.tan(x[w])))).all()


def test_device_core_on_host_warm_start_golden(host):
    """second control cycle (previous solution as the guess): tests/golden/carlike_min_time_n20_warm.npz"""
    g = np.load(os.path.join(GOLD, "carlike_min_time_n20_warm.npz"))
    cfg = A.config_carlike_min_time(20)
    xo, uo, do, st, it, kkt = host_solve(host, cfg, g["x0"], g["xf"], g["u_prev"], g["dt_prev"], init=(g["x_init"], g["u_init"], g["dt_init"]))
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(do - g["dt"]).max() < 1e-8
    assert (np.abs(it - g["iters"]) <= 2).all()


def test_log_of_frexp_mantissa_matches_libm(host):
    """LogAcc::value() evaluates log only on a frexp mantissa in [0.5, 1): mpc_core.hpp::log_mantissa (fdlibm kernel, <= 1 ulp)."""
    m = np.concatenate([np.linspace(0.5, 1.0, 200001)[:-1], [0.5, 0.70710678118654746, 0.7071067811865476, 0.9999999999999999]])
    out = np.empty_like(m)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    host.hostdbg_log_mantissa(C.c_int(m.size), p(m), p(out))
    assert np.abs(out - np.log(m)).max() < 2.3e-16
    bad = np.array([0.0, -0.6]); o2 = np.empty_like(bad)
    host.hostdbg_log_mantissa(C.c_int(2), p(bad), p(o2))
    assert np.isnan(o2).all()


def test_device_core_on_host_integral_form_golden(host):
    """integral-form quadratic cost on the fixed-dt grid = weights scaled by dt (mpc_problem.hpp); numpy-oracle fixture"""
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_integral_n20.npz"))
    cfg = A.config_unicycle_quadratic(20, integral_form=True)
    xo, uo, do, st, it, kkt = host_solve(host, cfg, g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6
    assert (np.abs(it - g["iters"]) <= 2).all()


def test_device_core_on_host_closed_loop_golden(host):
    """SURVEY 8c level 3: 40 closed-loop cycles of config 1 (fixed-dt grid, shifted warm start), re-solved independently."""
    g = np.load(os.path.join(GOLD, "unicycle_quadratic_closed_loop_n20.npz"))
    cfg = A.config_unicycle_quadratic(20)
    xo, uo, do, st, it, kkt = host_solve(host, cfg, g["x0"], g["xf"], g["u_prev"], g["dt_prev"], init=(g["x_init"], g["u_init"], g["dt_init"]))
    assert (st == 0).all()
    assert np.abs(xo - g["x"]).max() < 1e-6 and np.abs(uo - g["u"]).max() < 1e-6
    assert (np.abs(it - g["iters"]) <= 2).all()


def test_inertia_count_of_a_combine_block_on_the_host(host):
    """mpc_core.hpp::pit_block_inertia_tri -- the arithmetic the partitioned sweep's combine runs on the device to count the negative eigenvalues of its pivot block --
    compiled for the host: n-(W) + n-(P - W^-1) - 5 against numpy's eigenvalues on random symmetric pairs, definite and not; a vanishing pivot is reported, not guessed."""
    import ctypes as C
    rng = np.random.default_rng(5)
    host.host_pit_block_inertia.restype = C.c_int
    ok = C.c_int(0)
    seen = set()
    for t in range(400):
        A = rng.normal(size=(5, 5)); W = A + A.T
        if t % 3 == 0:
            W = -(A @ A.T) - 0.05 * np.eye(5)               # the usual case: W negative definite
        B = rng.normal(size=(5, 5)); P = B + B.T
        if t % 5 == 0:
            P[:2, :] = 0.0; P[:, :2] = 0.0                      # no curvature in the position rows (minimum time without clearance rows)
        r = host.host_pit_block_inertia(np.ascontiguousarray(W).ctypes.data_as(C.c_void_p), np.ascontiguousarray(P).ctypes.data_as(C.c_void_p), C.byref(ok))
        want = int((np.linalg.eigvalsh(W) < 0).sum() + (np.linalg.eigvalsh(P - np.linalg.inv(W)) < 0).sum()) - 5
        assert ok.value == 1 and r == want, (t, r, want)
        seen.add(want)
    assert {0, 1, 2}.issubset(seen) and min(seen) < 0
    Z = np.zeros((5, 5)); Z[0, 1] = Z[1, 0] = 1.0            # zero leading pivot
    host.host_pit_block_inertia(Z.ctypes.data_as(C.c_void_p), np.eye(5).ctypes.data_as(C.c_void_p), C.byref(ok))
    assert ok.value == 0


MODEL_PARAMS = {0: (), 1: (0.4,), 2: (0.4,), 3: (1.0, 1.3)}


@pytest.mark.parametrize("model", sorted(MODEL_PARAMS))
@pytest.mark.parametrize("method", [0, 1, 2])
def test_kernel_core_collocation_rows_equal_the_oracle_s_reference_form_rows_on_the_heading_manifold(host, model, method):
    """The kernel forms its rows as c = dt F(theta_1, u, dt) - (x_2 - x_1) with the midpoint / second Crank-Nicolson point placed by the explicit heading
    relation (mpc_core.hpp::model_trig_colloc) instead of interpolate_angle(theta_1, theta_2): the same thing wherever theta_2 satisfies the rule's own
    heading row -- so the comparison with the numpy oracle's REFERENCE-FORM rule (se2_nlp.collocation_defect: fd_collocation_se2.h:54-69, :91-108, :130-147 restated literally,
    the Crank-Nicolson aliasing included) is made there (theta_2 = theta_1 + dt f_2 for forward / midpoint differences, theta_1 + 2 dt f_2 for the literal
    Crank-Nicolson rule); positions are arbitrary.  dt * reference-form error == kernel row."""
    rng = np.random.default_rng(100 * model + method)
    N = 400
    par = MODEL_PARAMS[model]
    x1 = np.column_stack([rng.uniform(-5, 5, N), rng.uniform(-5, 5, N), rng.uniform(-np.pi, np.pi, N)])
    x1[:16, 2] = np.pi * np.array([1, -1] * 8) * (1 - 1e-16 * np.arange(16))            # headings at +-pi
    u = np.column_stack([rng.uniform(-0.2, 0.4, N), rng.uniform(-1.2, 1.2, N)])
    dt = rng.uniform(0.01, 1.0, N)
    f = np.array([R.dynamics(model, par, a, b) for a, b in zip(x1, u)])                  # the heading rate does not depend on the pose
    x2 = np.column_stack([rng.uniform(-5, 5, N), rng.uniform(-5, 5, N), np.zeros(N)])
    step = (2.0 if method == 2 else 1.0) * dt * f[:, 2]
    x2[:, 2] = [R.normalize_theta(a + b) for a, b in zip(x1[:, 2], step)]
    ref = np.stack([R.collocation_defect(method, model, par, x1[i:i + 1], u[i:i + 1], x2[i:i + 1], dt[i])[0] for i in range(N)]) * dt[:, None]
    cfg = A.make_config(model=model, model_params=par if par else (0.0,), n=20, collocation=method)
    c = np.zeros_like(x1)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    x1c, uc, x2c, dtc = (np.ascontiguousarray(a) for a in (x1, u, x2, dt))
    host.hostdbg_colloc(C.byref(cfg), C.c_int(N), p(x1c), p(uc), p(x2c), p(dtc), c.ctypes.data_as(C.c_void_p))
    assert np.abs(c - ref).max() < 5e-15, np.abs(c - ref).max()
    small = np.abs(step) < 3.0                                                            # (a heading step beyond pi wraps: there the row is 2 pi on BOTH sides)
    assert np.abs(c[small, 2]).max() < 5e-15                                              # on the heading manifold the heading row vanishes


def test_kernel_core_accept_step_is_the_oracle_s_retraction(host):
    """SURVEY.md 8 row a15 (VectorVertexSE2::plus, include/mpc_local_planner/optimal_control/vector_vertex_se2.h:79-96: add the increment, wrap the heading): the host build of
    the kernel core's accept step (x += alpha dx; heading = normalize_theta, mpc_wave.hpp::xt / accept) at alpha = 1 against the numpy restatement, bit for bit -- on vertices
    with the heading at +-pi and one ulp inside, and increments of 0, +-1e-17, +-1e-9, +-pi, 2 pi."""
    rng = np.random.default_rng(15)
    N = 400
    v = np.column_stack([rng.uniform(-10, 10, N), rng.uniform(-10, 10, N), rng.uniform(-np.pi, np.pi, N)])
    d = rng.uniform(-4, 4, (N, 3))
    v[:32, 2] = np.pi; v[32:64, 2] = np.nextafter(np.pi, 0) * np.array([1, -1] * 16)
    special = np.array([0.0, 1e-17, -1e-17, 1e-9, -1e-9, np.pi, -np.pi, 2 * np.pi])
    d[:64, 2] = np.tile(special, 8)
    want = v + d
    want[:, 2] = [R.normalize_theta(t) for t in want[:, 2]]
    assert ((want[:, 2] >= -np.pi) & (want[:, 2] < np.pi)).all()
    out = np.empty_like(v)
    host.hostdbg_retract.restype = None
    host.hostdbg_retract(C.c_int(N), np.ascontiguousarray(v).ctypes.data_as(C.c_void_p), np.ascontiguousarray(d).ctypes.data_as(C.c_void_p), C.c_double(1.0), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, want)
    from oracle import candidates as OC
    assert np.array_equal(OC.wrap(v[:, 2] + d[:, 2]), want[:, 2])
