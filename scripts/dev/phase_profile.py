"""developer tool: per-phase cycle counters of the solve kernel (build: _lib.build(extra_flags=("-DMPC_PROFILE=1",), out=".../libmpc_hip_prof.so"); run with MPC_HIP_LIB pointing at it)"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _lib
B, n = 1024, 50
s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
r = s.solve(x0, xf, up, dtp)
r = s.solve(x0, xf, up, dtp)
print("kernel ms", s.last_kernel_ms(), "converged", (r.status == 0).mean(), "iters", r.iters.mean())
lib = _lib.load()
buf = np.zeros((B, 16), dtype=np.int64)
lib.mpc_debug_profile(buf.ctypes.data_as(C.c_void_p), C.c_int(B))
names = ["ticks", "wall100MHz", "iters", "nfac", "ntrial", "kkt", "barrier_terms", "backward", "forward", "post", "logs0", "trial", "accept", "bwd_loop", "bwd_setup", "fwd_loop"]
print("tick rate GHz ~", (buf[:, 0] / (buf[:, 1] / 100e6)).mean() / 1e9)
tot = buf.sum(0).astype(float)
print("per iteration (ticks):", {k: round(tot[i] / tot[2]) for i, k in enumerate(names) if i not in (1, 2, 3, 4)})
print("factorisations per iteration", tot[3] / tot[2], "trials per iteration", tot[4] / tot[2])
print("per-sweep ticks: backward", tot[7] / tot[3], "forward", tot[8] / (tot[2]), "trial", tot[11] / max(1, tot[4]))
