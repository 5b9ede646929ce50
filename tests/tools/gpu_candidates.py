#!/usr/bin/env python
"""Developer script (GPU box): candidate initial trajectories on the device vs the same rule on the C oracle, config 2.
usage: python tests/tools/gpu_candidates.py [B] [caps e.g. 60,60,60]"""
import json, os, sys, time
import numpy as np
import torch
torch.cuda.init(); torch.zeros(1, device='cuda')      # torch's HIP runtime first (bench.py order)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A
from oracle import c_oracle as CO, se2_nlp as R, candidates as OC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
caps = tuple(int(c) for c in sys.argv[2].split(",")) if len(sys.argv) > 2 else (60, 60, 60)
kinds = (A.CAND_REFERENCE, A.CAND_BLEND, A.CAND_BLEND_REVERSE, A.CAND_TRAVEL)[:len(caps)]
n = 50
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
out = {}
for tag, kw in [("single", {}), ("cand", dict(candidates=kinds, candidate_max_iter=caps))]:
    cfg = m.config_carlike_min_time(n, **kw)
    s = m.BatchSolver(cfg, max_batch=B)
    r = s.solve(x0, xf, up, dtp)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = [T(a) for a in (x0, xf, up, dtp)]
    xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
    do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    ms = []
    for k in range(8):
        s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr())
        s.synchronize(); ms.append(s.last_kernel_ms())
    win, tot = s.last_candidates(B)
    assert (st.cpu().numpy() == r.status).all() and np.array_equal(xo.cpu().numpy(), r.x), "device-pointer and host-pointer entries disagree / not deterministic"
    ok = r.status == 0
    out[tag] = dict(kernel_ms=float(np.mean(ms[2:])), converged=float(ok.mean()), iters_mean=float(r.iters.mean()), iters_p99=float(np.percentile(r.iters, 99)),
                    iters_max=int(r.iters.max()), iters_total_mean=float(tot.mean()), winners=np.bincount(win + 1, minlength=len(caps) + 1).tolist(),
                    solves_per_s=B / np.mean(ms[2:]) * 1e3, converged_solves_per_s=ok.sum() / np.mean(ms[2:]) * 1e3)
    print(tag, json.dumps(out[tag]))
    if tag == "cand":
        t = time.time()
        ocfg = R.config_carlike_min_time(n)
        ox, ou, od, ost, oit, owin, olow, allr = OC.solve_candidates(CO, lambda cap: CO.from_nlp_config(ocfg, max_iter=cap), x0, xf, up, dtp, kinds, caps, n, ocfg.dt_ref)
        same_w = win == owin
        both = ok & (ost == 0) & same_w
        err = np.abs(r.x - ox).reshape(B, -1).max(1)
        print("oracle rule: %.1f s; converged %.4f; winners equal %.4f; of those both converged %d: median |dx| %.2e, <1e-6 %.4f, <1e-4 %.4f; device total iters >= oracle lower bound: %s"
              % (time.time() - t, (ost == 0).mean(), same_w.mean(), both.sum(), np.median(err[both]), (err[both] < 1e-6).mean(), (err[both] < 1e-4).mean(), bool((tot >= olow - 2).all())))
        print("winner mismatch detail (gpu, oracle):", [(int(a), int(b)) for a, b in zip(win[~same_w], owin[~same_w])][:20])
    s.close()
json.dump(out, open(os.path.join("gpurun_out", "candidates_B%d.json" % B), "w"), indent=1)
