#!/usr/bin/env python
"""Developer tool (GPU box): per-phase cycle counters of the solve kernel on the config-3 workload (unicycle n = 80, 16 polygons).
Needs a library built with -DMPC_PROFILE=1 -DMPC_DEV_ONE_MODEL=0:  MPC_HIP_LIB=<that .so> python scripts/gpu_phase_profile_config3.py"""
import sys, os, ctypes as C, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _lib
n, B, O, V, M = 80, 2048, 16, 6, 4
x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
for tag, kw, ob in (("with obstacles", dict(max_obstacles=O, max_vertices=V, max_obstacle_rows=M), obs), ("without", {}, None)):
    s = m.BatchSolver(m.config_unicycle_quadratic(n, **kw), max_batch=B)
    r = s.solve(x0, xf, up, dtp, obstacles=ob)
    r = s.solve(x0, xf, up, dtp, obstacles=ob)
    print(tag, "kernel ms", s.last_kernel_ms(), "iters", r.iters.mean(), "conv", (r.status == 0).mean())
    lib = _lib.load()
    buf = np.zeros((B, 16), dtype=np.int64)
    lib.mpc_debug_profile(buf.ctypes.data_as(C.c_void_p), C.c_int(B))
    names = ["ticks", "wall100MHz", "iters", "nfac", "ntrial", "kkt", "barrier_terms", "backward", "forward", "post", "logs0", "trial", "accept", "bwd_loop", "bwd_setup", "fwd_loop"]
    it = buf[:, 2].sum()
    print("  per iteration ticks:", {k: round(float(buf[:, i].sum() / it)) for i, k in enumerate(names) if i >= 5 or i == 0}, "fac/it", round(buf[:, 3].sum() / it, 3), "trials/it", round(buf[:, 4].sum() / it, 3))
    s.close()
