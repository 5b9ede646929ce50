"""developer script (GPU): find an instance whose iteration count differs between the partitioned and the serial sweeps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
B, n = 64, 50
inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
out = {}
for no_pit in (True, False):
    if no_pit: os.environ["MPC_NO_PIT"] = "1"
    else: os.environ.pop("MPC_NO_PIT", None)
    s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
    out[no_pit] = s.solve(*inp)
    s.close()
a, b = out[True], out[False]
d = np.nonzero(a.iters != b.iters)[0]
print("differing instances", d.tolist())
print("serial iters", a.iters[d].tolist()); print("pit iters   ", b.iters[d].tolist())
print("serial status", a.status[d].tolist()); print("pit status   ", b.status[d].tolist())
