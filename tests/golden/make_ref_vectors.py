#!/usr/bin/env python
"""Generates tests/golden/ref_models_collocation.npz: outputs of REFERENCE code -- include/mpc_local_planner/utils/math_utils.h, the four robot
models of include/mpc_local_planner/systems/ and the three collocation rules of include/mpc_local_planner/optimal_control/fd_collocation_se2.h --
compiled from /root/reference into oracle/_ref/libmpc_ref.so (`make -C oracle ref`; oracle/ref_wrap.cpp says what is real and what is an
interface stand-in) on seeded inputs.  /root/reference does not exist on the GPU box, so the vectors are committed; this script must be run
where the reference tree is.   usage: python tests/golden/make_ref_vectors.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_lib as RL      # noqa: E402

MODELS = {0: (), 1: (0.4,), 2: (0.4,), 3: (1.0, 1.3)}      # ids of include/mpc_hip.h; parameters: wheelbase | (lr, lf)


def inputs(seed=20260925, K=400):
    rng = np.random.default_rng(seed)
    pi = np.pi
    th = np.concatenate([rng.uniform(-25, 25, K), [pi, -pi, 3 * pi, -3 * pi, 2 * pi, 0.0, np.nextafter(pi, 0), np.nextafter(-pi, 0), 1e-300, -1e-300, 7 * pi, -7 * pi, 100.0, -100.0]])
    a1, a2, fr = rng.uniform(-4, 4, K), rng.uniform(-4, 4, K), rng.uniform(0, 1, K)
    avg = rng.uniform(-pi, pi, (40, 7))
    x1 = rng.uniform(-3, 3, (K, 3)); x1[:, 2] = rng.uniform(-pi, pi, K)
    x2 = x1 + rng.uniform(-0.5, 0.5, (K, 3)); x2[:, 2] = RL.normalize_theta(x1[:, 2] + rng.uniform(-1.2, 1.2, K))
    x2[::7, 2] = RL.normalize_theta(x1[::7, 2] + rng.uniform(2.5, 3.7, x2[::7].shape[0]))        # heading differences across the +-pi seam
    u = np.stack([rng.uniform(-0.5, 0.8, K), rng.uniform(-1.2, 1.2, K)], 1)
    dt = rng.uniform(0.02, 0.6, K)
    return th, a1, a2, fr, avg, x1, x2, u, dt


def main():
    assert RL.build(), "oracle/_ref cannot be built here (no /root/reference)"
    th, a1, a2, fr, avg, x1, x2, u, dt = inputs()
    out = dict(theta=th, normalize_theta=RL.normalize_theta(th), a1=a1, a2=a2, factor=fr, interpolate_angle=RL.interpolate_angle(a1, a2, fr),
               angle_sets=avg, average_angles=np.array([RL.average_angles(r) for r in avg]), x1=x1, x2=x2, u=u, dt=dt)
    for mid, par in MODELS.items():
        out[f"dynamics_model{mid}"] = RL.dynamics(mid, par, x1, u)
        f = out[f"dynamics_model{mid}"]
        for method in (0, 1, 2):
            out[f"collocation_model{mid}_method{method}"] = RL.collocation(method, mid, par, x1, u, x2, dt)
            # the same with the second heading ON the rule's heading manifold (theta_2 = theta_1 + dt f_2; literal Crank-Nicolson: + 2 dt f_2), where the
            # kernel's explicit-heading formulation of the row coincides with the reference's (tests/test_reference_pinned.py)
            th2 = RL.normalize_theta(x1[:, 2] + (2.0 if method == 2 else 1.0) * dt * f[:, 2])
            x2m = x2.copy(); x2m[:, 2] = th2
            out[f"manifold_theta2_model{mid}_method{method}"] = th2
            out[f"manifold_collocation_model{mid}_method{method}"] = RL.collocation(method, mid, par, x1, u, x2m, dt)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_models_collocation.npz"), **out)
    print("written", len(out), "arrays")


if __name__ == "__main__":
    main()
