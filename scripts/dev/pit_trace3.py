import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["MPC_HIP_LIB"] = os.path.join(root, "mpc_local_planner_amd", "csrc", "libmpc_hip_nancheck.so")
if sys.argv[1] == "serial": os.environ["MPC_NO_PIT"] = "1"
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
B, n = 64, 50
inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
r = s.solve(*inp)
print("iters", r.iters[20])
