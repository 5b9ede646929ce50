"""developer script: statuses / iteration counts of the one-candidate config-2 batch against the C solver's (computed on the CPU beforehand: python tests/tools/dev/pivot_ab.py cpu)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import mpc_local_planner_amd as m
F = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pivot_ab_oracle.npz")
B = 1024
inp = m.workloads.carlike_min_time_inputs(B, seed=20260924)
cfg = m.config_carlike_min_time(50)
if sys.argv[1:] == ["cpu"]:
    from oracle import c_oracle
    o = c_oracle.solve_batch(c_oracle.from_nlp_config(cfg), *inp)
    np.savez(F, status=o[3], iters=o[4], x=o[0]); print(np.bincount(o[3])); sys.exit()
o = np.load(F)
s = m.BatchSolver(cfg, max_batch=B); r = s.solve(*inp); s.close()
print("device", np.bincount(r.status, minlength=4), "oracle", np.bincount(o["status"], minlength=4), "same status", (r.status == o["status"]).sum(), "same iters", (r.iters == o["iters"]).sum())
d = np.nonzero(r.status != o["status"])[0]
print([(int(i), int(r.status[i]), int(r.iters[i]), int(o["status"][i]), int(o["iters"][i])) for i in d])
