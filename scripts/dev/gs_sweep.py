"""developer check (GPU): global form == LDS form bit for bit (fp64) over a sweep of grid sizes around the row pitch's / the partitioned sweeps' boundaries, with and without
clearance rows, and with ragged grids.  Under MPC_POISON_GSTAGE=1 the pool of global blocks starts as NaN patterns: a word consumed before it is written would show up here."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A
bad = 0
def cmp(label, mk, inp, B, **kw):
    global bad
    out = {}
    for mode in (A.STAGE_LDS, A.STAGE_GLOBAL):
        try:
            s = m.BatchSolver(mk(mode), max_batch=B)
        except Exception as e:
            print(label, "mode", mode, "refused:", str(e)[:80]); return
        if "n_grid" in kw: s.set_grid_sizes(kw["n_grid"])
        out[mode] = s.solve(*inp, **{k: v for k, v in kw.items() if k != "n_grid"}); s.close()
    a, g = out[A.STAGE_LDS], out[A.STAGE_GLOBAL]
    same = all(np.array_equal(getattr(a, f), getattr(g, f), equal_nan=True) for f in ("x", "u", "dt", "status", "iters"))
    print(f"{label}: {'identical' if same else 'DIFFERENT'}  converged {np.mean(a.status == 0):.3f} iters {a.iters.mean():.1f}", flush=True)
    bad += 0 if same else 1
B = 64
for n in (12, 20, 39, 40, 41, 47, 48, 49, 63, 64, 65, 80, 96, 112, 127, 128, 129, 160, 200):
    cmp(f"car-like n={n}", lambda mode: m.config_carlike_min_time(n, stage_data=mode), m.workloads.carlike_min_time_inputs(B, seed=n), B)
for n in (30, 48, 64, 80, 100):
    x0, xf, up, dtp, obstacles = m.workloads.unicycle_obstacle_inputs(B, n_obst=8, max_vertices=5, lateral=(0.15, 0.8))
    cmp(f"unicycle + polygons n={n}", lambda mode: m.config_unicycle_quadratic(n, max_obstacles=8, max_vertices=5, max_obstacle_rows=3, max_iter=60, stage_data=mode), (x0, xf, up, dtp), B, obstacles=obstacles)
for n in (50, 90):
    rng = np.random.default_rng(n)
    ng = rng.integers(8, n + 1, B).astype(np.int32)
    cmp(f"bicycle ragged n<= {n}", lambda mode: m.config_bicycle_min_time(n, stage_data=mode), m.workloads.bicycle_min_time_inputs(B), B, n_grid=ng)
print("DIFFERENCES" if bad else "ALL IDENTICAL")
