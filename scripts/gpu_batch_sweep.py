"""Device-resident throughput of the headline workload (car-like min-time, n=50, cold start) versus the batch size on one GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
dev = torch.device("cuda", 0)
n = 50
for B in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    a, b, c, d = T(x0), T(xf), T(up), T(dtp)
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    cand = dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)) if "--single" not in sys.argv else {}
    s = m.BatchSolver(m.config_carlike_min_time(n, **cand), max_batch=B)
    ms = []
    for rep in range(3):
        s.solve_device(B, a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize(); ms.append(s.last_kernel_ms())
    k = min(ms[1:])
    cv = float((st == 0).float().mean())
    print(f"B={B:6d}: kernel {k:8.2f} ms  {B * cv / k * 1e3:9.0f} converged solves/s   converged {cv:.4f}  iterations of all candidates per instance {s.last_candidates(B)[1].mean():.1f}")
    s.close()
