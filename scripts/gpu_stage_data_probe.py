"""Where a solve keeps its factorisation data: everything in LDS (MPC_STAGE_LDS) against stage records + gains in global memory (MPC_STAGE_GLOBAL,
mpc_wave.hpp::GlobalStage) on the workloads whose LDS record keeps SIMDs empty -- results compared bit for bit, kernel time, LDS bytes, workgroups per CU.
    python scripts/gpu_stage_data_probe.py [quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401  (HIP runtime first)
import mpc_local_planner_amd as m  # noqa: E402
from mpc_local_planner_amd import _abi as A  # noqa: E402

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
C5 = dict(candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0))
C2 = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 60, 50, 40), candidate_param=(0.0, 2.0, 3.0, 1.5))


def run(tag, mk_cfg, B, inputs, obstacles=None, steps=3):
    out = {}
    for mode, name in ((A.STAGE_LDS, "lds"), (A.STAGE_GLOBAL, "global"), (A.STAGE_AUTO, "auto")):
        s = m.BatchSolver(mk_cfg(stage_data=mode), max_batch=B)
        r = s.solve(*inputs, obstacles=obstacles)
        ms = []
        for _ in range(steps):
            t0 = time.perf_counter()
            r = s.solve(*inputs, obstacles=obstacles)
            ms.append((time.perf_counter() - t0) * 1e3)
        kms = s.last_kernel_ms()
        lds = s.lds_bytes()
        out[name] = (r, kms, lds)
        print(f"[{tag}] {name:6s}: kernel {kms:8.3f} ms  (host-pointer call {min(ms):8.3f} ms)  LDS {lds:6d} B -> {(160 * 1024) // lds if lds else 0} per CU (register cap 4)  "
              f"converged {np.mean(r.status == 0):.4f}  iters mean {r.iters.mean():.2f}  -> {B * np.mean(r.status == 0) / (kms * 1e-3) / 1e3:8.1f} k converged solves/s", flush=True)
        s.close()
    a, b = out["lds"][0], out["global"][0]
    same = all(np.array_equal(getattr(a, f), getattr(b, f)) for f in ("x", "u", "dt", "status", "iters"))
    print(f"[{tag}] global == lds bit for bit: {same}   speed-up of the kernel: x{out['lds'][1] / out['global'][1]:.2f}\n", flush=True)
    return same


ok = True
n5, B5 = 120, 1024
inp5 = m.workloads.bicycle_min_time_inputs(B5)
ok &= run("config 5 shape, fp64, candidates", lambda **k: m.config_bicycle_min_time(n5, precision=A.FP64, **C5, **k), B5, inp5)
ok &= run("config 5 shape, fp64, reference path", lambda **k: m.config_bicycle_min_time(n5, precision=A.FP64, **k), B5, inp5)
ok &= run("config 5 shape, MPC_MIXED, candidates", lambda **k: m.config_bicycle_min_time(n5, precision=A.MIXED, **C5, **k), B5, inp5)
ok &= run("config 5 shape, fp32 tol 1e-4, candidates", lambda **k: m.config_bicycle_min_time(n5, precision=A.FP32, tol=1e-4, **C5, **k), B5, inp5)
if not quick:
    n3, B3, O, V, M = 80, 4096, 16, 6, 4
    x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(B3, n_obst=O, max_vertices=V, lateral=(0.15, 0.8))
    ok &= run("config 3, rows binding", lambda **k: m.config_unicycle_quadratic(n3, max_obstacles=O, max_vertices=V, max_obstacle_rows=M, max_iter=60, **k), B3, (x0, xf, up, dtp), obstacles=obs)
    B4 = 4096
    ok &= run("config 4 share (n = 50, B = 4096), candidates", lambda **k: m.config_carlike_min_time(50, **C2, **k), B4, m.workloads.carlike_min_time_inputs(B4))
    B8 = 8192
    ok &= run("config 5 shape, fp64, candidates, B = 8192", lambda **k: m.config_bicycle_min_time(n5, precision=A.FP64, **C5, **k), B8, m.workloads.bicycle_min_time_inputs(B8), steps=1)
print("ALL BIT-IDENTICAL" if ok else "DIFFERENCES FOUND")
sys.exit(0 if ok else 1)
