import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/tools: algo_stats.py
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import workloads as W
CO.build()
B=1024
x0,xf,up,dtp=W.carlike_min_time_inputs(B)
oc=CO.from_nlp_config(R.config_carlike_min_time(50),max_iter=100)
import algo_stats
algo_stats.set_algo(sys.argv[2:])
xo,uo,do,st,it=CO.solve_batch(oc,x0,xf,up,dtp)
order=np.argsort(-it)
sel = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv)>1 and sys.argv[1]!="-" else order[:3].tolist()
lib=CO._load()
for i in sel:
    print("=== instance",i,"status",st[i],"iters",it[i], "x0",x0[i],"xf",xf[i])
    lib.oracle_set_trace(1)
    CO.solve_batch(oc,x0[i:i+1],xf[i:i+1],up[i:i+1],dtp[i:i+1],nthreads=1)
    lib.oracle_set_trace(0)
