#!/usr/bin/env python
"""Generates tests/golden/cold_start_scipy_config{2,3}.npz: an INDEPENDENT solver (scipy SLSQP, an active-set SQP with no code in common
with the interior-point implementations of this repository) on the REFERENCE-FORM NLP (oracle/se2_nlp.py::ReferenceNlp), started from the
reference's cold start -- not from anybody's answer (level 2).

  config 2   car-like minimum time, n = 50, the first 32 instances of the SURVEY 8d distribution
  config 5   (shape of BASELINE configs[4]) kinematic bicycle minimum time, n = 120, the first 8 instances of workloads.bicycle_min_time_inputs (r06)
  config 3   unicycle quadratic form, n = 80, 16 polygon obstacles, the first 32 instances of workloads.unicycle_obstacle_inputs
             (association frozen on the cold start, max 4 rows per grid point, as the batched solvers do)

Start point: the 2-pose-plan cold start (src/controller.cpp:807-857 + full_discretization_grid_base_se2.cpp:192-239) with the controls
seeded from the state guess exactly as the interior-point solvers do (oracle/ipm_dense.py::controls_from_states; with u = 0 the car-like
model has no steering authority and SLSQP's first QP is singular in the steering direction).

Stored per instance: SLSQP's x, u, dt, objective, success flag, iterations and its max constraint violation.  tests/test_oracle_solver.py
compares the C oracle with it on the CPU, tests/test_gpu_parity.py the device.  Runtime: ~2 min (config 2) + ~25 min (config 3) on 8 cores.
usage: python tests/golden/make_cold_start_scipy.py [2|3|3b|3c|5] [count]"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import se2_nlp as R, ipm_dense as I, kkt_check as KC      # noqa: E402
import mpc_local_planner_amd.workloads as W                           # noqa: E402  (numpy only)


def solve_one(args):
    cfgname, i, x0, xf, up, dtp, obst = args
    t0 = time.time()
    if cfgname == 2:
        ocfg = R.config_carlike_min_time(50)
        inp = R.CycleInputs(x0=x0, xf=xf, u_prev=up, dt_prev=float(dtp))
        nlp = R.ReferenceNlp(ocfg, inp)
    elif cfgname == 5:
        ocfg = R.config_bicycle_min_time(120)
        inp = R.CycleInputs(x0=x0, xf=xf, u_prev=up, dt_prev=float(dtp))
        nlp = R.ReferenceNlp(ocfg, inp)
    else:
        ocfg = R.config_unicycle_quadratic(80)
        obs = KC.obstacle_list(*obst)
        inp = R.CycleInputs(x0=x0, xf=xf, u_prev=up, dt_prev=float(dtp), obstacles=obs)
        rel, rel_dyn = R.associate_obstacles(ocfg, R.cold_start(ocfg, x0, xf), obs, 4)
        nlp = R.ReferenceNlp(ocfg, inp, relevant=rel, relevant_dyn=rel_dyn)
    start = I.controls_from_states(ocfg, R.cold_start(ocfg, x0, xf))
    z0 = nlp.pack(start)
    lb, ub = nlp.bounds()
    z0 = np.minimum(np.maximum(z0, lb), ub)
    bnds = [(None if l < -1e29 else l, None if u > 1e29 else u) for l, u in zip(lb, ub)]
    if ocfg.dt_free:
        bnds[-1] = (1e-3, ocfg.dt_ub)            # the reference rows divide by dt
    cons = [{"type": "eq", "fun": nlp.equalities}, {"type": "ineq", "fun": lambda z: -nlp.inequalities(z)}]
    r = minimize(nlp.objective, z0, method="SLSQP", bounds=bnds, constraints=cons, options=dict(maxiter=600, ftol=1e-12))
    t = nlp.unpack(r.x)
    viol = max(np.abs(nlp.equalities(r.x)).max(), nlp.inequalities(r.x).max(initial=0.0))
    print(f"config {cfgname} instance {i}: success {r.success} it {r.nit} f {r.fun:.6f} viol {viol:.1e} ({time.time() - t0:.0f} s)", flush=True)
    return t.x, t.u, t.dt, r.fun, bool(r.success), r.nit, viol


def main():
    arg = sys.argv[1] if len(sys.argv) > 1 else "2"
    # 3b / 3c (r04): config 3 on placements where the clearance rows BIND -- polygons 0.15 .. 0.8 m beside the start-goal line (the bench leg's
    # placement: rows start violated by up to 5 cm) and 0.02 .. 0.6 m (polygons reach to within 2 cm of the line: rows start violated by up to 18 cm, detours)
    lateral = {"3b": (0.15, 0.8), "3c": (0.02, 0.6)}.get(arg)
    suffix = {"3b": "_binding", "3c": "_touching"}.get(arg, "")
    which = int(arg[0])
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    if which == 2:
        x0, xf, up, dtp = W.carlike_min_time_inputs(K)
        jobs = [(2, i, x0[i], xf[i], up[i], dtp[i], None) for i in range(K)]
    elif which == 5:          # (r06) config-5 shape: kinematic bicycle, n = 120, goals 5 .. 40 m; ~1.5 min per instance, so 8 instances by default
        K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        x0, xf, up, dtp = W.bicycle_min_time_inputs(K)
        jobs = [(5, i, x0[i], xf[i], up[i], dtp[i], None) for i in range(K)]
    else:
        x0, xf, up, dtp, (no, nv, vt) = W.unicycle_obstacle_inputs(K, n_obst=16, max_vertices=6, **(dict(lateral=lateral) if lateral else {}))
        jobs = [(3, i, x0[i], xf[i], up[i], dtp[i], (no[i], nv[i], vt[i], None, None)) for i in range(K)]
    with mp.Pool(min(K, os.cpu_count() or 2)) as pool:
        out = pool.map(solve_one, jobs)
    n = {2: 50, 5: 120}.get(which, 80)
    u = np.zeros((K, n, 2))
    for i, o in enumerate(out):
        u[i, :n - 1] = o[1]; u[i, n - 1] = o[1][-1]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"cold_start_scipy_config{which}{suffix}.npz"),
                        x=np.stack([o[0] for o in out]), u=u, dt=np.array([o[2] for o in out]), objective=np.array([o[3] for o in out]),
                        success=np.array([o[4] for o in out]), nit=np.array([o[5] for o in out]), violation=np.array([o[6] for o in out]),
                        count=K, lateral=np.array(lateral if lateral else (0.3, 1.5)),
                        generator=json.dumps(dict(script="tests/golden/make_cold_start_scipy.py", solver="scipy.optimize.minimize(method='SLSQP', maxiter=600, ftol=1e-12) on oracle/se2_nlp.py::ReferenceNlp",
                                                  start="reference cold start, controls seeded from the state guess (oracle/ipm_dense.py::controls_from_states)"), sort_keys=True))


if __name__ == "__main__":
    main()
