#!/usr/bin/env python
"""Developer tool (GPU box): per-phase cycle counters of the solve kernel.  Needs a library built with -DMPC_PROFILE=1 (and, for a quick
build, -DMPC_DEV_ONE_MODEL): MPC_HIP_LIB=<that .so> python scripts/gpu_phase_profile.py"""
import sys, os, ctypes as C, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _lib
B, n = 1024, 50
s = m.BatchSolver(m.config_carlike_min_time(n, max_iter=60), max_batch=B)
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
r = s.solve(x0, xf, up, dtp)
r = s.solve(x0, xf, up, dtp)
print("kernel ms", s.last_kernel_ms())
lib = _lib.load()
buf = np.zeros((B, 16), dtype=np.int64)
lib.mpc_debug_profile(buf.ctypes.data_as(C.c_void_p), C.c_int(B))
names = ["ticks", "wall100MHz", "iters", "nfac", "ntrial", "kkt", "barrier_terms", "backward", "forward", "post", "logs0", "trial", "accept", "bwd_loop", "bwd_setup", "fwd_loop"]
print("tick rate GHz ~", (buf[:, 0] / (buf[:, 1] / 100e6)).mean() / 1e9)
tot = buf[:, 0].sum()
print("share of all ticks:", {k: round(float(buf[:, i].sum() / tot), 4) for i, k in enumerate(names) if i >= 5})
it = buf[:, 2].sum()
print("per iteration ticks:", {k: round(float(buf[:, i].sum() / it)) for i, k in enumerate(names) if i >= 5 or i == 0}, "fac/it", buf[:, 3].sum() / it, "trials/it", buf[:, 4].sum() / it)
print("per-sweep ticks: backward", buf[:, 7].sum() / buf[:, 3].sum(), "forward", buf[:, 8].sum() / buf[:, 3].sum(), "post", buf[:, 9].sum() / buf[:, 3].sum(), "trial", buf[:, 11].sum() / max(1, buf[:, 4].sum()))
