"""One workload, a few launches of the solve kernel: the command rocprofv3 wraps for the per-leg profiles (scripts/profile_cmd.sh).
    python scripts/gpu_solve_once.py <config5_fp64|config5_fp64_lds|config3|config3_lds|config4|headline|n20_one_wave|n20_two_waves> [launches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
import mpc_local_planner_amd as m  # noqa: E402
from mpc_local_planner_amd import _abi as A  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "config5_fp64"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = A.STAGE_LDS if what.endswith("_lds") else A.STAGE_AUTO
what = what[:-4] if what.endswith("_lds") else what
obstacles = None
if what == "config5_fp64":
    B = 1024
    cfg = m.config_bicycle_min_time(120, candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0), stage_data=mode)
    inp = m.workloads.bicycle_min_time_inputs(B)
elif what == "config3":
    B = 4096
    x0, xf, up, dtp, obstacles = m.workloads.unicycle_obstacle_inputs(B, n_obst=16, max_vertices=6, lateral=(0.15, 0.8))
    inp = (x0, xf, up, dtp)
    cfg = m.config_unicycle_quadratic(80, max_obstacles=16, max_vertices=6, max_obstacle_rows=4, max_iter=60, stage_data=mode)
elif what in ("n20_one_wave", "n20_two_waves"):      # the reference's shipped grid size at saturation, reference path alone: one wave per SIMD against mpc_config.two_wave_min_batch's kernel
    B = 32768
    cfg = m.config_carlike_min_time(20, two_wave_min_batch=-1 if what == "n20_one_wave" else 0)
    inp = m.workloads.carlike_min_time_inputs(B, goal_range=(1.0, 2.4))
elif what == "config4":
    B = 4096
    cfg = m.config_carlike_min_time(50, candidates=(0, 5, 5, 7), candidate_max_iter=(100, 60, 50, 40), candidate_param=(0.0, 2.0, 3.0, 1.5), stage_data=mode)
    inp = m.workloads.carlike_min_time_inputs(B)
else:
    B = 1024
    cfg = m.config_carlike_min_time(50, candidates=(0, 5, 5, 7), candidate_max_iter=(100, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5), stage_data=mode)
    inp = m.workloads.carlike_min_time_inputs(B)
s = m.BatchSolver(cfg, max_batch=B)
for _ in range(reps):
    r = s.solve(*inp, obstacles=obstacles)
win, tot = s.last_candidates(B)
print(f"{what} stage_data={'lds' if mode == A.STAGE_LDS else 'auto'}: B {B} kernel {s.last_kernel_ms():.3f} ms LDS {s.lds_bytes()} B converged {np.mean(r.status == 0):.4f} "
      f"iterations winner {r.iters.mean():.2f} all candidates {tot.mean():.2f}")
s.close()
