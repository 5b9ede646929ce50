"""developer tool: per-phase cycle counters of the solve kernel (build: _lib.build(extra_flags=("-DMPC_PROFILE=1",), out=".../libmpc_hip_prof.so"); run with MPC_HIP_LIB pointing at it)
    python scripts/dev/phase_profile.py [carlike50|bicycle120|unicycle80] [lds|global|auto] [B]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _lib, _abi as A
what = sys.argv[1] if len(sys.argv) > 1 else "carlike50"
mode = {"lds": A.STAGE_LDS, "global": A.STAGE_GLOBAL, "auto": A.STAGE_AUTO}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
obstacles = None
if what == "carlike50":
    cfg = m.config_carlike_min_time(50, stage_data=mode); inp = m.workloads.carlike_min_time_inputs(B)
elif what == "bicycle120":
    cfg = m.config_bicycle_min_time(120, stage_data=mode); inp = m.workloads.bicycle_min_time_inputs(B)
else:
    x0, xf, up, dtp, obstacles = m.workloads.unicycle_obstacle_inputs(B, n_obst=16, max_vertices=6, lateral=(0.15, 0.8)); inp = (x0, xf, up, dtp)
    cfg = m.config_unicycle_quadratic(80, max_obstacles=16, max_vertices=6, max_obstacle_rows=4, max_iter=60, stage_data=mode)
s = m.BatchSolver(cfg, max_batch=B)
r = s.solve(*inp, obstacles=obstacles)
r = s.solve(*inp, obstacles=obstacles)
print(what, sys.argv[2:] , "kernel ms", s.last_kernel_ms(), "lds", s.lds_bytes(), "converged", (r.status == 0).mean(), "iters", r.iters.mean())
lib = _lib.load()
R = min(B, 4096)
buf = np.zeros((R, 16), dtype=np.int64)
lib.mpc_debug_profile(buf.ctypes.data_as(C.c_void_p), C.c_int(R))
names = ["ticks", "wall100MHz", "iters", "nfac", "ntrial", "kkt", "barrier_terms", "backward", "forward", "post", "logs0", "trial", "accept", "bwd_loop", "bwd_setup", "fwd_loop"]
print("tick rate GHz ~", (buf[:, 0] / (buf[:, 1] / 100e6)).mean() / 1e9)
tot = buf.sum(0).astype(float)
print("per iteration (ticks):", {k: round(tot[i] / tot[2]) for i, k in enumerate(names) if i not in (1, 2, 3, 4)})
print("factorisations per iteration", tot[3] / tot[2], "trials per iteration", tot[4] / tot[2])
print("per-sweep ticks: backward", tot[7] / tot[3], "forward", tot[8] / (tot[2]), "trial", tot[11] / max(1, tot[4]))
