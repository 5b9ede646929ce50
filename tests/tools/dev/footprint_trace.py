"""developer script: per-iteration trace of ONE instance of a workload of tests/tools/dev/footprint_mismatch.py
    python tests/tools/dev/footprint_trace.py cpu "static line" 102      -- the C oracle's (oracle_set_trace)
    MPC_HIP_LIB=<a -DMPC_NANCHECK=102 single-TU build> python tests/tools/dev/footprint_trace.py gpu "static line" 102"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import footprint_mismatch as FM
mode, label, inst = args[0], args[1], int(args[2])
fn, k = [(f, kk) for l, f, kk in FM.CASES if l == label][0]
B, n, inp, obs, kw, (kind, params, dmin, dyn, O) = fn(k)
if mode == "cpu":
    from oracle import c_oracle, se2_nlp as R
    ocfg = R.config_carlike_min_time(n)
    ocfg.footprint_kind, ocfg.footprint_params, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = kind, params, dmin, 0.5, 2.5
    if dyn: ocfg.enable_dynamic_obstacles = True
    sl = slice(inst, inst + 1)
    lib = c_oracle._load(); lib.oracle_set_trace(1)
    ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *[a[sl] for a in inp], obstacles=tuple(a[sl] for a in obs), obst=c_oracle.obst_from_nlp_config(ocfg, O, 1, 4), nthreads=1)
    print("oracle status", ref[3], "iters", ref[4])
else:
    import mpc_local_planner_amd as m
    s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
    r = s.solve(*inp, obstacles=obs); s.close()
    print("device status", r.status[inst], "iters", r.iters[inst])
