"""PINNED to executed reference code: the angle helpers of include/mpc_local_planner/utils/math_utils.h (:36-103) -- the one part of the reference that compiles in this image
from its own sources (oracle/ref_math.cpp includes it from /root/reference; nothing is copied, no stand-in header is involved).  tests/golden/ref_math_utils.npz holds what it
computes (generator: tests/golden/make_ref_math_vectors.py).  Held to it bit for bit: the numpy oracle, the oracle's candidate guesses, the HOST BUILD of the kernel core
(csrc/mpc_core.hpp -- the source the HIP kernel compiles) and the C++ facade.  Everything else of the reference's hot path includes Eigen / corbo / ROS / teb headers, which the
image lacks: the oracle's restatements of those files are UNPINNED.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import se2_nlp as R
from oracle import ref_math as RM

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_math_utils.npz"))


@pytest.mark.skipif(not os.path.isdir(RM.REFERENCE_INCLUDE), reason="the reference tree is only present in the build container")
def test_recorded_vectors_are_what_the_compiled_reference_gives_today():
    assert RM.build()
    assert np.array_equal(RM.normalize_theta(G["theta"]), G["normalize_theta"])
    assert np.array_equal(RM.interpolate_angle(G["a1"], G["a2"], G["factor"]), G["interpolate_angle"])
    assert np.array_equal(RM.cross2d(G["v1"], G["v2"]), G["cross2d"])
    dt, ds = RM.distance_points2d(G["v1"], G["v2"])
    assert np.array_equal(dt, G["distance_templated"]) and np.array_equal(ds, G["distance_scalar"])
    assert np.array_equal([RM.average_angles(G[f"set{i}"]) for i in range(int(G["n_sets"]))], G["average_angles"])
    # fresh inputs, the compiled header against the numpy restatement
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    t = rng.uniform(-50, 50, 2000)
    assert np.array_equal(RM.normalize_theta(t), [R.normalize_theta(x) for x in t])
    a, b, f = rng.uniform(-np.pi, np.pi, 500), rng.uniform(-np.pi, np.pi, 500), rng.uniform(-1, 3, 500)
    assert np.array_equal(RM.interpolate_angle(a, b, f), [R.interpolate_angle(x, y, z) for x, y, z in zip(a, b, f)])


def test_angle_helpers_reproduce_the_reference_bit_for_bit():
    ours = np.array([R.normalize_theta(t) for t in G["theta"]])
    assert np.array_equal(ours, G["normalize_theta"])
    assert ((G["normalize_theta"] >= -np.pi) & (G["normalize_theta"] < np.pi)).all()          # [-pi, pi): +pi maps to -pi
    ours = np.array([R.interpolate_angle(a, b, f) for a, b, f in zip(G["a1"], G["a2"], G["factor"])])
    assert np.array_equal(ours, G["interpolate_angle"])
    from oracle import candidates as OC
    assert np.array_equal(OC.wrap(G["theta"]), G["normalize_theta"])
    assert np.array_equal([R.cross2d(a, b) for a, b in zip(G["v1"], G["v2"])], G["cross2d"])
    assert np.array_equal(G["distance_templated"], G["distance_scalar"])
    assert np.array_equal(np.sqrt((G["v2"][:, 0] - G["v1"][:, 0]) ** 2 + (G["v2"][:, 1] - G["v1"][:, 1]) ** 2), G["distance_scalar"])


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
    """csrc/mpc_core.hpp::normalize_theta, compiled for the host from the source the HIP kernel compiles"""
    ours = np.array([host.hostdbg_normalize_theta(float(t)) for t in G["theta"]])
    assert np.array_equal(ours, G["normalize_theta"])


def test_facade_interpolate_angle_reproduces_the_reference():
    """include/mpc_controller.hpp::interpolate_angle (the facade's resampling / warm-start extrapolation)"""
    src = os.path.join(HERE, "host_harness", "controller_host.cpp")
    out = os.path.join(HERE, "host_harness", "_build", "libctl_host_pin.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wl,--unresolved-symbols=ignore-all", src, "-o", out], check=True)
    lib = C.CDLL(out)
    lib.ctl_interpolate_angle.restype = C.c_double
    lib.ctl_interpolate_angle.argtypes = [C.c_double] * 3
    ours = np.array([lib.ctl_interpolate_angle(float(a), float(b), float(f)) for a, b, f in zip(G["a1"], G["a2"], G["factor"])])
    assert np.array_equal(ours, G["interpolate_angle"])
