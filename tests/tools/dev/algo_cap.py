import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/tools: algo_stats.py
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import workloads as W
import algo_stats
algo_stats.set_algo(sys.argv[1:])
x0,xf,up,dtp=W.carlike_min_time_inputs(1024)
oc=CO.from_nlp_config(R.config_carlike_min_time(50),max_iter=400)
xo,uo,do,st,it=CO.solve_batch(oc,x0,xf,up,dtp)
print("cap 400: conv",(st==0).mean(), "hist of iters:", np.histogram(it[st==0],bins=[0,15,20,25,30,40,50,75,100,150,200,300,400])[0].tolist(), "status", np.bincount(st,minlength=5).tolist())
