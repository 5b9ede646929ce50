"""which instances differ between MPC_STAGE_LDS and MPC_STAGE_GLOBAL in fp32, and is either mode run-to-run deterministic?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A
C5 = dict(candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0))
B = 1024
inp = m.workloads.bicycle_min_time_inputs(B)
def go(tag, **kw):
    rs = {}
    for mode, name in ((A.STAGE_LDS, "lds"), (A.STAGE_GLOBAL, "glb")):
        for rep in range(2):
            s = m.BatchSolver(m.config_bicycle_min_time(120, stage_data=mode, **kw), max_batch=B)
            r1 = s.solve(*inp); w1 = s.last_candidates(B)[0].copy()
            r2 = s.solve(*inp); w2 = s.last_candidates(B)[0].copy()
            s.close()
            rs[(name, rep, 0)] = (r1, w1); rs[(name, rep, 1)] = (r2, w2)
    ref = rs[("lds", 0, 0)]
    for k, (r, w) in rs.items():
        d = np.flatnonzero((r.status != ref[0].status) | (r.iters != ref[0].iters) | (np.abs(r.x - ref[0].x).reshape(B, -1).max(1) > 0))
        print(f"[{tag}] {k}: {len(d)} instances differ from lds/0/0", [(int(i), int(ref[0].status[i]), int(r.status[i]), int(ref[0].iters[i]), int(r.iters[i]), int(ref[1][i]), int(w[i])) for i in d[:8]], flush=True)
go("fp32 cand", precision=A.FP32, tol=1e-4, **C5)
go("fp32 single", precision=A.FP32, tol=1e-4)
go("fp32 single tol 1e-8-ish", precision=A.FP32, tol=1e-6)
go("fp64 cand", precision=A.FP64, **C5)
