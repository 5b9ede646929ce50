"""developer tool: where the slowest reference-path solves of the config-2 batch spend their time (MPC_HIP_LIB = a -DMPC_PROFILE single-TU build, see phase_profile.py)"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _lib
B = 1024
inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=B)
r = s.solve(*inp); r = s.solve(*inp)
print("kernel ms", s.last_kernel_ms(), "status", np.bincount(r.status, minlength=4))
buf = np.zeros((B, 16), dtype=np.int64)
_lib.load().mpc_debug_profile(buf.ctypes.data_as(C.c_void_p), C.c_int(B))
names = ["ticks", "wall100MHz", "iters", "nfac", "ntrial", "kkt", "barrier_terms", "backward", "forward", "post", "logs0", "trial", "accept", "bwd_loop", "bwd_setup", "fwd_loop"]
order = np.argsort(-buf[:, 0])
print("slowest 12 (ticks, us, iters, nfac, ntrial, status):")
for i in order[:12]:
    b = buf[i]; print(int(i), int(b[0]), round(b[1] / 100.0, 1), int(b[2]), int(b[3]), int(b[4]), int(r.status[i]), {k: int(b[j]) for j, k in enumerate(names) if j >= 5 and j <= 12})
slow = r.iters >= 100
for nm, sel in (("100-iteration solves", slow), ("the rest", ~slow)):
    t = buf[sel].sum(0).astype(float)
    print(nm, int(sel.sum()), "ticks/iter", round(t[0] / t[2]), "fac/iter", round(t[3] / t[2], 2), "trials/iter", round(t[4] / t[2], 2), {k: round(t[j] / t[2]) for j, k in enumerate(names) if 5 <= j <= 12})
