"""developer script (GPU): the two-waves-per-SIMD kernel against the one-wave kernels of the same handle configuration, over the batch size.  mpc_config.two_wave_min_batch = -1 switches the two-wave
kernel off, 1 on for every launch; W2_N=<grid points> (default 50: there only a library built with -DMPC_DEV_SWITCHES has a two-wave kernel -- the global form, MPC_W2_GS=1).  Prints kernel times, converged solves/s and whether the outputs are
bit-identical."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m

def run(cfg, inp, B, reps=4):
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(*inp)
    ms = []
    for _ in range(reps):
        r = s.solve(*inp); ms.append(s.last_kernel_ms())
    lds = s.lds_bytes()
    s.close()
    return min(ms), r

sets = {"headline100": dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)),
        "share100_60_50_40": dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 60, 50, 40), candidate_param=(0.0, 2.0, 3.0, 1.5)),
        "single": {}}
N = int(os.environ.get("W2_N", "50"))
which = sys.argv[1:] or ["headline100", "share100_60_50_40"]
for name in which:
    print(f"# {name}, n = {N}")
    for B in (1024, 2048, 4096, 8192, 32768):
        inp = m.workloads.carlike_min_time_inputs(B, goal_range=(1.0, 6.0 * N / 50.0))
        t1, r1 = run(m.config_carlike_min_time(N, two_wave_min_batch=-1, **sets[name]), inp, B)
        t2, r2 = run(m.config_carlike_min_time(N, two_wave_min_batch=1, **sets[name]), inp, B)
        same = np.array_equal(r1.x, r2.x) and np.array_equal(r1.u, r2.u) and np.array_equal(r1.dt, r2.dt) and np.array_equal(r1.status, r2.status) and np.array_equal(r1.iters, r2.iters)
        cv = np.mean(r1.status == 0)
        print(f"B={B:6d}: one wave {t1:8.3f} ms {B * cv / t1:8.1f} k/s | W2 {t2:8.3f} ms {B * cv / t2:8.1f} k/s | x{t1 / t2:.3f} | bit-identical {same} | converged {cv:.4f}", flush=True)
