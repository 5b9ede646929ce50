"""Generator of tests/golden/ref_math_utils.npz: inputs and outputs of the reference's utils/math_utils.h (include/mpc_local_planner/utils/math_utils.h:36-103), EXECUTED -- it is
the one part of the reference that compiles in this image from its own sources (oracle/ref_math.cpp, `make -C oracle ref`).  Run where /root/reference exists:
    python tests/golden/make_ref_math_vectors.py
The GPU box has no reference tree: the device tests compare against these vectors."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_math as RM      # noqa: E402


def inputs():
    rng = np.random.default_rng(20260930)
    pi = np.pi
    edge = [0.0, -0.0, pi, -pi, np.nextafter(pi, 0), np.nextafter(-pi, 0), np.nextafter(pi, 4), np.nextafter(-pi, -4), 2 * pi, -2 * pi, 3 * pi, -3 * pi, 1e-300, -1e-300,
            1e6, -1e6, 1e15 + 0.5, 6.283185307179586, 12.566370614359172, 0.5 * pi, -0.5 * pi, 7.0, -7.0, 100.0, 3.2, -3.2]
    theta = np.concatenate([edge, rng.uniform(-pi, pi, 200), rng.uniform(-40, 40, 400), rng.uniform(-1e4, 1e4, 100)])
    a1 = np.concatenate([rng.uniform(-pi, pi, 300), [pi, -pi, 3.1, -3.1, 0.0, 3.0]])
    a2 = np.concatenate([rng.uniform(-pi, pi, 300), [-pi, pi, -3.1, 3.1, 0.0, -3.0]])
    factor = np.concatenate([rng.uniform(0, 1, 200), rng.uniform(-1, 3, 100), [0.5, 0.5, 0.5, 2.0, 0.0, 1.0]])      # 2.0: the extrapolation of warmStartShifting (full_discretization_grid_base_se2.cpp:287-298)
    sets = [rng.uniform(-pi, pi, k) for k in (1, 2, 3, 5, 8, 13)] + [np.array([0.0, pi]), np.array([0.3, -0.3]), np.array([pi, -pi])]
    v1 = rng.uniform(-5, 5, (300, 2)); v2 = rng.uniform(-5, 5, (300, 2))
    return theta, a1, a2, factor, sets, v1, v2


if __name__ == "__main__":
    assert RM.build(), "needs /root/reference (make -C oracle ref)"
    theta, a1, a2, factor, sets, v1, v2 = inputs()
    dt, ds = RM.distance_points2d(v1, v2)
    out = dict(theta=theta, normalize_theta=RM.normalize_theta(theta), a1=a1, a2=a2, factor=factor, interpolate_angle=RM.interpolate_angle(a1, a2, factor),
               v1=v1, v2=v2, cross2d=RM.cross2d(v1, v2), distance_templated=dt, distance_scalar=ds,
               average_angles=np.array([RM.average_angles(s) for s in sets]), n_sets=np.array(len(sets)),
               generator=json.dumps(dict(script="tests/golden/make_ref_math_vectors.py", source="/root/reference/mpc_local_planner/include/mpc_local_planner/utils/math_utils.h compiled by oracle/ref_math.cpp (g++ -O2 -ffp-contract=off)")))
    for i, s in enumerate(sets):
        out[f"set{i}"] = s
    np.savez(os.path.join(ROOT, "tests", "golden", "ref_math_utils.npz"), **out)
    print("wrote tests/golden/ref_math_utils.npz:", {k: np.shape(v) for k, v in out.items() if not k.startswith("set")})
