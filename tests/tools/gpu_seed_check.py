#!/usr/bin/env python
"""Developer script (GPU box): the bench's candidate set on EIGHT seeds of the config-2 distribution (the set was chosen with the C oracle on these)."""
import os, sys
import numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
B, n = 1024, 50
s = m.BatchSolver(m.config_carlike_min_time(n, candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)), max_batch=B)
dev = torch.device("cuda", 0)
for k in range(8):
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=m.workloads.SEED_CONFIG2 + k)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = [T(a) for a in (x0, xf, up, dtp)]
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    ms = []
    for r in range(4):
        s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize(); ms.append(s.last_kernel_ms())
    ok = (st == 0).float().mean().item()
    print(f"seed +{k}: converged {ok:.4f}  kernel {np.mean(ms[1:]):.3f} ms  converged solves/s {B * ok / np.mean(ms[1:]) * 1e3:.0f}  winner iterations p99 {np.percentile(it.cpu().numpy()[st.cpu().numpy() == 0], 99):.0f}")
