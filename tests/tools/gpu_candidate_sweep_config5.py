#!/usr/bin/env python
"""Developer script (GPU box): kernel time / converged fraction of the config-5 share (kinematic bicycle, n = 120, fp32, tol 1e-4, B = 1024)
over candidate sets and iteration caps, measured in the arithmetic the leg runs in (a set chosen with the fp64 oracle was worse in fp32)."""
import json, os, sys
import numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 120
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # "fp32" (tol 1e-4: the leg of rounds 2-4) or "fp64" (tol 1e-8, global form: the leg that counts since r05)
x0, xf, up, dtp = m.workloads.bicycle_min_time_inputs(B)
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = [T(a) for a in (x0, xf, up, dtp)]
xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
P = (0, 2.0, 3.0, 1.5)
CASES = [((0,), (100,), ()), ((0, 5, 5, 7), (100, 100, 100, 100), P),        # the reference solve alone; the headline's set at caps 100 (what the leg ran before)
         ((0, 1, 5, 7), (80, 70, 60, 50), (0, 0, 2.0, 1.5)), ((0, 1, 4, 5), (80, 70, 60, 50), (0, 0, 0, 2.0)), ((0, 1, 2, 5), (80, 70, 60, 50), (0, 0, 0, 2.0)),
         ((0, 1, 2, 5), (70, 60, 50, 45), (0, 0, 0, 2.0)), ((0, 1, 2, 5), (60, 50, 45, 40), (0, 0, 0, 2.0)), ((0, 1, 2, 5), (50, 50, 45, 40), (0, 0, 0, 2.0)),
         ((0, 1, 2), (70, 60, 50), ()), ((1, 0, 5, 7), (80, 70, 60, 50), (0, 0, 2.0, 1.5)),
         ((0, 1, 2, 5), (55, 45, 40, 35), (0, 0, 0, 2.0)), ((0, 1, 2, 5), (50, 45, 40, 35), (0, 0, 0, 2.0)), ((0, 1, 2, 5), (45, 45, 40, 35), (0, 0, 0, 2.0)), ((0, 1, 2, 5), (60, 60, 60, 60), (0, 0, 0, 2.0)),
         ((0, 1, 2, 5, 7), (60, 50, 45, 40, 40), (0, 0, 0, 2.0, 1.5)), ((0, 1, 2, 5), (100, 60, 50, 45), (0, 0, 0, 2.0))]
SEEDS = [None, 1, 2, 3, 4, 5]       # None = the workload's own seed; the others: the chosen set (0,1,2,5) (60,50,45,40) on other draws
def run(kinds, caps, par, tag=None):
    kw = dict(candidates=kinds, candidate_max_iter=caps, candidate_param=par) if len(kinds) > 1 else dict(max_iter=caps[0])
    pk = dict(precision=1, tol=1e-4) if PREC == "fp32" else dict(precision=0)
    s = m.BatchSolver(m.config_bicycle_min_time(n, **pk, **kw), max_batch=B)
    ms = []
    for k in range(4):
        s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize(); ms.append(s.last_kernel_ms())
    win, tot = s.last_candidates(B)
    ok = st.cpu().numpy() == 0
    itn = it.cpu().numpy()
    print(json.dumps(dict(seed=tag, kinds=kinds, caps=caps, param=par, kernel_ms=round(float(np.mean(ms[1:])), 3), converged=round(float(ok.mean()), 4), iters_total=round(float(tot.mean()), 1),
                          iters_winner_mean=round(float(itn[ok].mean()), 1), iters_winner_p99=float(np.percentile(itn[ok], 99)),
                          conv_solves_per_s=round(ok.sum() / np.mean(ms[1:]) * 1e3), winners=np.bincount(win + 1, minlength=len(kinds) + 1).tolist())), flush=True)
    s.close()

for kinds, caps, par in CASES:
    run(kinds, caps, par)
for sd in SEEDS[1:]:
    x0, xf, up, dtp = m.workloads.bicycle_min_time_inputs(B, seed=sd)
    for i, a in enumerate((x0, xf, up, dtp)): d[i].copy_(T(a))
    run((0, 1, 2, 5), (60, 50, 45, 40), (0, 0, 0, 2.0), tag=sd)
