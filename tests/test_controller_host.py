"""CPU test of the C++ Controller facade's host logic (include/mpc_controller.hpp) against the oracle's restatement
of Controller::generateInitialStateTrajectory + initializeSequences(xinit)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import se2_nlp as R

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "controller_host.cpp")
OUT = os.path.join(HERE, "host_harness", "_build", "libctl_host.so")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # the facade only needs the ABI's declarations here: link lazily, the tested functions make no library calls
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wl,--unresolved-symbols=ignore-all", SRC, "-o", OUT], check=True)
    l = C.CDLL(OUT)
    l.ctl_interpolate_angle.restype = C.c_double
    l.ctl_interpolate_angle.argtypes = [C.c_double] * 3
    l.ctl_resample.restype = C.c_double
    l.ctl_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int]
    return l


def test_initial_state_trajectory_matches_oracle(lib):
    rng = np.random.default_rng(4)
    n, dt_ref = 20, 0.3
    for P in (2, 3, 7):
        plan = np.cumsum(rng.uniform(0.1, 0.6, (P, 3)), axis=0)
        plan[:, 2] = rng.uniform(-3, 3, P)
        x0 = plan[0].copy(); xf = plan[-1].copy()
        for est in (True, False):
            out = np.zeros((n, 3))
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            lib.ctl_initial_state_trajectory(C.c_int(P), p(plan), p(x0), p(xf), C.c_int(n), C.c_double(dt_ref), C.c_int(est), p(out))
            t, v = R.generate_initial_state_trajectory(plan, x0, xf, n, dt_ref, est)
            cfg = R.OcpConfig(n=n, dt_ref=dt_ref)
            ref = R.initialize_sequences_xinit(cfg, x0, xf, t, v)
            np.testing.assert_allclose(out, ref.x, atol=1e-14)
    # 2-pose plan == the device-side cold start
    x0 = np.array([0.0, 0.0, 3.0]); xf = np.array([1.0, 2.0, -3.0])
    t, v = R.generate_initial_state_trajectory(np.stack([x0, xf]), x0, xf, n, dt_ref)
    a = R.initialize_sequences_xinit(R.OcpConfig(n=n, dt_ref=dt_ref), x0, xf, t, v)
    b = R.cold_start(R.OcpConfig(n=n, dt_ref=dt_ref), x0, xf)
    np.testing.assert_allclose(a.x, b.x, atol=1e-15)
    assert lib.ctl_interpolate_angle(3.0, -3.0, 0.5) == pytest.approx(float(R.interpolate_angle(3.0, -3.0, 0.5)))


def test_resample_matches_oracle(lib):
    # FullDiscretizationGridBaseSE2::resampleTrajectory, src/optimal_control/full_discretization_grid_base_se2.cpp:440-524
    rng = np.random.default_rng(9)
    for n, n_new in ((12, 13), (12, 11), (5, 4), (20, 21)):
        x = np.cumsum(rng.uniform(0.0, 0.3, (n, 3)), axis=0); x[:, 2] = rng.uniform(-3.1, 3.1, n)
        u = rng.uniform(-0.2, 0.4, (n - 1, 2))
        tr = R.resample_trajectory(R.Trajectory(x.copy(), u.copy(), 0.27), n_new)
        cap = max(n, n_new)
        xb = np.zeros((cap, 3)); ub = np.zeros((cap, 2))
        xb[:n] = x; ub[:n - 1] = u; ub[n - 1] = u[-1]
        dt_new = lib.ctl_resample(xb.ctypes.data_as(C.c_void_p), ub.ctypes.data_as(C.c_void_p), 0.27, n, n_new)
        assert dt_new == pytest.approx(tr.dt)
        np.testing.assert_allclose(xb[:n_new], tr.x, atol=1e-14)
        np.testing.assert_allclose(ub[:n_new - 1], tr.u, atol=1e-14)


def test_warm_start_shifting_matches_oracle(lib):
    """fixed-grid moving-horizon warm start (full_discretization_grid_base_se2.cpp:241-339) vs oracle/se2_nlp.py"""
    rng = np.random.default_rng(9)
    n = 20
    for trial in range(20):
        x = np.cumsum(rng.uniform(0.0, 0.2, (n, 3)), axis=0)
        x[:, 2] = np.cumsum(rng.uniform(-0.5, 0.5, n))
        x[:, 2] = (x[:, 2] + np.pi) % (2 * np.pi) - np.pi
        u = rng.normal(size=(n - 1, 2))
        k = int(rng.integers(0, 6))
        x0 = x[k] + rng.normal(scale=0.01, size=3) * (trial % 3 > 0)
        ref = R.warm_start_shifting(R.Trajectory(x.copy(), u.copy(), 0.3), x0)
        xx = x.copy(); uu = np.vstack([u, u[-1:]]).copy()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        x0c = np.ascontiguousarray(x0)
        assert lib.ctl_find_nearest_state(p(xx), C.c_int(n), p(x0c)) == R.find_nearest_state(R.Trajectory(x, u, 0.3), x0)
        lib.ctl_warm_start_shifting(p(xx), p(uu), C.c_int(n), p(x0c))
        assert np.abs(xx - ref.x).max() < 1e-14
        assert np.abs(uu[:-1] - ref.u).max() < 1e-14


def test_optimal_control_result_wire_layout(lib):
    """f4: the facade's OptimalControlResult against the numpy restatement of msg/OptimalControlResult.msg + src/controller.cpp:197-221, bit-exact."""
    rng = np.random.default_rng(9)
    n, dt = 17, 0.2371
    x = rng.normal(size=(n, 3)); u = rng.normal(size=(n, 2)); u[-1] = u[-2]
    out = np.zeros(9 + 2 * n + 5 * n)
    lib.ctl_optimal_control_result.restype = C.c_int
    cnt = lib.ctl_optimal_control_result(C.c_int(n), x.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), C.c_double(dt), C.c_int(1), C.c_double(0.0123), C.c_int(41),
                                         out.ctypes.data_as(C.c_void_p))
    ref = R.optimal_control_result(x, u, dt, True, 0.0123, 41)
    assert cnt == out.size
    assert out[:9].tolist() == [41, 3, 2, 1, 0.0123, n, 3 * n, n, 2 * n]
    np.testing.assert_array_equal(out[9:9 + n], ref["time_states"])
    np.testing.assert_array_equal(out[9 + n:9 + 4 * n], ref["states"])
    np.testing.assert_array_equal(out[9 + 4 * n:9 + 5 * n], ref["time_controls"])
    np.testing.assert_array_equal(out[9 + 5 * n:], ref["controls"])
    assert ref["states"][3 * 5 + 2] == x[5, 2] and ref["controls"][2 * 7 + 1] == u[7, 1]      # sample-major ("column major" of the dim x N matrix)
