"""The experiment matrix of DESIGN.md section 3.1 on the config-5 SHAPE (kinematic bicycle, n = 120, goals 5 .. 40 m; BASELINE configs[4]), reference path alone, C oracle
(VERDICT r05 item 3): which instances fail within the reference's 100 iterations, how they end, and what Ipopt's machinery (filter + second-order corrections, probing
oracle, kkt-error globalisation, convexified fallback / Hessian, the other barrier rule, other barrier starts, more iterations) changes.
    B=1024 python tests/tools/algo_stats_config5.py          (profiles/r06_config5_algo_matrix.log)"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import workloads as W

KEYS = dict(mu=0, glob=1, soc=2, safeguard=3, sigma_max=4, fix_fact=5, cf=6, sr1=7, sreset=8, inertia=9, elastic=10, etrig=11)
lib = CO._load()
B = int(os.environ.get("B", 1024))
inp = W.bicycle_min_time_inputs(B)
cfg = R.config_bicycle_min_time(120)
STATUS = ["converged", "iteration limit", "line search", "factorisation", "numerical", "time limit"]


def setk(**kw):
    for k, v in kw.items():
        lib.oracle_set_algo(C.c_int(KEYS[k]), C.c_double(float(v)))


def run(tag, cap=100, **okw):
    oc = CO.from_nlp_config(cfg, max_iter=cap, **okw)
    t = time.time(); xo, uo, do, st, it = CO.solve_batch(oc, *inp); el = time.time() - t
    ok = st == 0
    hist = ", ".join(f"{STATUS[i]} {c}" for i, c in enumerate(np.bincount(st, minlength=6)) if c)
    print(f"{tag:52s} converged {ok.mean() * 100:5.1f} %   iterations of the converged: mean {it[ok].mean():5.1f} p90 {np.percentile(it[ok], 90):4.0f}   [{hist}]   ({el:.0f} s)", flush=True)
    return st, it, do


if __name__ == "__main__":
    print(f"# config-5 shape, {B} instances (workloads.bicycle_min_time_inputs), reference path alone (cold start of Controller::step, one candidate), C oracle, tol 1e-8")
    st, it, do = run("the algorithm (adaptive mu, l1 merit, inertia test), 100")
    run("... 150 iterations", 150)
    run("... 300 iterations", 300)
    setk(glob=1, soc=4); run("filter line search + 4 second-order corrections"); setk(glob=0, soc=0)
    setk(mu=1); run("Mehrotra probing oracle (mu_oracle = probing)"); setk(mu=0)
    setk(safeguard=1); run("adaptive_mu_globalization = kkt-error"); setk(safeguard=0)
    setk(cf=1); run("convexified fallback before delta_w"); setk(cf=0)
    setk(inertia=0); run("inertia-free curvature test (r01-r03)"); setk(inertia=1)
    run("mu_strategy = monotone", mu_strategy=1)
    run("hessian_mode = convexified (limited-memory's stand-in)", hessian_mode=1)
    run("mu_init = 1", mu_init=1.0)
    run("mu_init = 0.01", mu_init=0.01)
    r = np.hypot(inp[1][:, 0], inp[1][:, 1])
    print("# by goal range (the algorithm, 100 iterations): " + ", ".join(f"{lo}-{lo + 5} m {np.mean(st[(r >= lo) & (r < lo + 5)] == 0) * 100:.0f} %" for lo in range(5, 40, 5)))
