"""developer script (GPU): partitioned vs serial sweeps.  usage: python scripts/dev/pit_check.py [debug]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1] if len(sys.argv) > 1 else "ab"
if mode == "debug":
    os.environ["MPC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "mpc_local_planner_amd", "csrc", "libmpc_hip_pitcheck.so")
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m

def run(B, n, no_pit, cands=False, seed=20260924, model="car"):
    if no_pit: os.environ["MPC_NO_PIT"] = "1"
    else: os.environ.pop("MPC_NO_PIT", None)
    kw = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)) if cands else {}
    if model == "car":
        cfg = m.config_carlike_min_time(n, **kw); inp = m.workloads.carlike_min_time_inputs(B, seed=seed)
    else:
        cfg = m.config_bicycle_min_time(n, **kw); inp = m.workloads.bicycle_min_time_inputs(B)
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(*inp)
    ms = []
    for _ in range(5):
        r = s.solve(*inp); ms.append(s.last_kernel_ms())
    s.close()
    return r, min(ms)

if mode == "debug":
    r, ms = run(64, 50, False)
    print("status", r.status, "iters", r.iters)
else:
    for (B, n, cands, model) in ((1024, 50, False, "car"), (1024, 50, True, "car"), (4096, 50, True, "car"), (256, 120, False, "bic"), (256, 81, False, "car"), (64, 40, False, "car"), (64, 43, False, "car")):
        a, ta = run(B, n, True, cands, model=model)
        b, tb = run(B, n, False, cands, model=model)
        both = (a.status == 0) & (b.status == 0)
        err = np.abs(a.x - b.x).reshape(B, -1).max(1)
        print(f"{model} B={B} n={n} cands={cands}: serial {ta:.3f} ms conv {np.mean(a.status == 0):.4f} iters {a.iters.mean():.2f} | pit {tb:.3f} ms conv {np.mean(b.status == 0):.4f} iters {b.iters.mean():.2f} | "
              f"same status {np.mean(a.status == b.status):.4f} same iters {np.mean(a.iters == b.iters):.4f} max|dx| over both-converged: median {np.median(err[both]):.2e} p99 {np.quantile(err[both], 0.99):.2e} max {err[both].max():.2e} frac<1e-6 {np.mean(err[both] < 1e-6):.4f}", flush=True)
