#!/usr/bin/env python
"""Developer script (GPU box): kernel time / converged fraction of config 2 (B = 1024) over candidate sets and iteration caps."""
import json, os, sys
import numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_local_planner_amd as m

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 50
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = [T(a) for a in (x0, xf, up, dtp)]
xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
CASES = [((0, 3, 4), (60, 60, 60), 8), ((0, 3, 4), (60, 50, 45), 8), ((0, 3, 4, 1), (60, 55, 45, 40), 8), ((0, 3, 4, 2), (60, 55, 45, 35), 8), ((0, 3, 4, 3), (60, 55, 45, 40), 8),
         ((0, 3, 4, 1), (55, 55, 50, 45), 8), ((0, 3, 4, 1), (60, 60, 50, 45), 8), ((0, 3, 4), (60, 60, 60), 12), ((0, 3, 4, 4), (60, 55, 45, 40), 12),
         ((0, 3, 4, 1), (50, 50, 50, 50), 8), ((0, 3, 4, 1), (60, 60, 60, 60), 8), ((3, 0, 4), (60, 60, 60), 8), ((3, 4, 0, 1), (55, 50, 45, 40), 8)]
for kinds, caps, blend in CASES:
    s = m.BatchSolver(m.config_carlike_min_time(n, candidates=kinds, candidate_max_iter=caps, candidate_blend=blend), max_batch=B)
    ms = []
    for k in range(6):
        s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize(); ms.append(s.last_kernel_ms())
    win, tot = s.last_candidates(B)
    ok = st.cpu().numpy() == 0
    print(json.dumps(dict(kinds=kinds, caps=caps, blend=blend, kernel_ms=round(float(np.mean(ms[2:])), 3), converged=round(float(ok.mean()), 4), iters_total=round(float(tot.mean()), 1),
                          conv_solves_per_s=round(ok.sum() / np.mean(ms[2:]) * 1e3), winners=np.bincount(win + 1, minlength=len(kinds) + 1).tolist())), flush=True)
    s.close()
