"""developer script (GPU): kernel times of a few workloads with the library named by MPC_HIP_LIB (run once per library); prints a checksum so that two libraries can be compared"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
def run(label, cfg, inp, B, reps=6):
    s = m.BatchSolver(cfg, max_batch=B)
    try:
        r = s.solve(*inp)
    except Exception as e:      # a one-model developer library
        print(f"{label:44s} skipped ({type(e).__name__})"); s.close(); return
    ms = []
    for _ in range(reps):
        r = s.solve(*inp); ms.append(s.last_kernel_ms())
    s.close()
    print(f"{label:44s} kernel {min(ms):8.3f} ms (median {np.median(ms):8.3f})  converged {np.mean(r.status == 0):.4f}  checksum {float(np.nansum(r.x[r.status == 0])):.9f}", flush=True)
LS = dict(line_search=int(os.environ["LS"])) if "LS" in os.environ else {}      # enum mpc_line_search (libraries from r06 on)
C = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5), **LS)
CL = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 60, 50, 40), candidate_param=(0.0, 2.0, 3.0, 1.5), **LS)
print("library:", os.environ.get("MPC_HIP_LIB", "product"), "line_search:", LS)
run("headline n50 B1024 c4", m.config_carlike_min_time(50, **C), m.workloads.carlike_min_time_inputs(1024), 1024)
run("single n50 B1024", m.config_carlike_min_time(50, **LS), m.workloads.carlike_min_time_inputs(1024), 1024)
run("config4 share n50 B4096 c4", m.config_carlike_min_time(50, **CL), m.workloads.carlike_min_time_inputs(4096), 4096)
run("n50 B32768 single", m.config_carlike_min_time(50, **LS), m.workloads.carlike_min_time_inputs(32768), 32768, reps=3)
run("config5 fp64 bicycle n120 B1024 c4", m.config_bicycle_min_time(120, candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0), **LS), m.workloads.bicycle_min_time_inputs(1024), 1024, reps=4)
run("n20 B32768 single one wave", m.config_carlike_min_time(20, two_wave_min_batch=-1, **LS), m.workloads.carlike_min_time_inputs(32768, goal_range=(1.0, 2.4)), 32768, reps=3)
run("n20 B32768 single two waves", m.config_carlike_min_time(20, two_wave_min_batch=1, **LS), m.workloads.carlike_min_time_inputs(32768, goal_range=(1.0, 2.4)), 32768, reps=3)
run("n24 B32768 single two waves", m.config_carlike_min_time(24, two_wave_min_batch=1, **LS), m.workloads.carlike_min_time_inputs(32768, goal_range=(1.0, 2.9)), 32768, reps=3)
run("n24 B8192 single two waves", m.config_carlike_min_time(24, two_wave_min_batch=1, **LS), m.workloads.carlike_min_time_inputs(8192, goal_range=(1.0, 2.9)), 8192, reps=3)
run("n24 B8192 single one wave", m.config_carlike_min_time(24, two_wave_min_batch=-1, **LS), m.workloads.carlike_min_time_inputs(8192, goal_range=(1.0, 2.9)), 8192, reps=3)
run("n24 B4096 single two waves", m.config_carlike_min_time(24, two_wave_min_batch=1, **LS), m.workloads.carlike_min_time_inputs(4096, goal_range=(1.0, 2.9)), 4096, reps=3)
run("n24 B4096 single one wave", m.config_carlike_min_time(24, two_wave_min_batch=-1, **LS), m.workloads.carlike_min_time_inputs(4096, goal_range=(1.0, 2.9)), 4096, reps=3)
