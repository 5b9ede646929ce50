"""Developer script (GPU box): config-5 share (bicycle n = 120, fp64, B = 1024) over the candidates' iteration caps."""
import json, os, sys
import numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
B, n = int(os.environ.get("B", 1024)), 120
inp = m.workloads.bicycle_min_time_inputs(B)
KINDS, PAR = (0, 1, 2, 5), (0.0, 0.0, 0.0, 2.0)
for caps in [(60, 50, 45, 40), (50, 50, 45, 40), (70, 50, 45, 40), (80, 50, 45, 40), (100, 50, 45, 40), (60, 60, 50, 40), (60, 45, 40, 35), (60, 40, 40, 40), (55, 45, 40, 35), (60, 55, 50, 45), (65, 50, 45, 40)]:
    s = m.BatchSolver(m.config_bicycle_min_time(n, candidates=KINDS, candidate_max_iter=caps, candidate_param=PAR), max_batch=B)
    ms = []
    for k in range(5):
        r = s.solve(*inp); ms.append(s.last_kernel_ms())
    win, tot = s.last_candidates(B)
    ok = r.status == 0
    print(json.dumps(dict(caps=caps, kernel_ms=round(float(min(ms[1:])), 3), converged=round(float(ok.mean()), 4), iters_total=round(float(tot.mean()), 1), conv_solves_per_s=round(ok.sum() / min(ms[1:]) * 1e3),
                          winners=np.bincount(win + 1, minlength=5).tolist())), flush=True)
    s.close()
