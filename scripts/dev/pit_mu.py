import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
def run(B, n, env, cands):
    for k in ("MPC_NO_PIT", "MPC_PIT_MU"): os.environ.pop(k, None)
    os.environ.update(env)
    kw = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)) if cands else {}
    s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
    inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
    r = s.solve(*inp); ms = []
    for _ in range(5):
        r = s.solve(*inp); ms.append(s.last_kernel_ms())
    s.close()
    return r, min(ms)
for cands in (False, True):
    base, tb = run(1024, 50, {"MPC_NO_PIT": "1"}, cands)
    print(f"cands={cands} serial: {tb:.3f} ms conv {np.mean(base.status == 0):.4f} iters {base.iters.mean():.2f}")
    for mu in ("0", "1e-8", "1e-6", "1e-5", "1e-4", "1e-3"):
        r, t = run(1024, 50, {"MPC_PIT_MU": mu}, cands)
        both = (r.status == 0) & (base.status == 0)
        err = np.abs(r.x - base.x).reshape(1024, -1).max(1)
        print(f"   pit while mu > {mu}: {t:.3f} ms conv {np.mean(r.status == 0):.4f} iters {r.iters.mean():.2f} same iters {np.mean(r.iters == base.iters):.4f} same status {np.mean(r.status == base.status):.4f} frac(|dx|<1e-6) {np.mean(err[both] < 1e-6):.4f}", flush=True)
