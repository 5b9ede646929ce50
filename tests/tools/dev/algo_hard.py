import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/tools: algo_stats.py
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import workloads as W
import algo_stats
algo_stats.set_algo(sys.argv[1:])
B=1024
x0,xf,up,dtp=W.carlike_min_time_inputs(B)
oc=CO.from_nlp_config(R.config_carlike_min_time(50),max_iter=100)
xo,uo,do,st,it=CO.solve_batch(oc,x0,xf,up,dtp)
r=np.hypot(xf[:,0],xf[:,1]); bearing=np.arctan2(xf[:,1],xf[:,0])
wrap=lambda a:(a+np.pi)%(2*np.pi)-np.pi
rel_b=np.abs(wrap(bearing-x0[:,2])); rel_y=np.abs(wrap(xf[:,2]-x0[:,2])); rel_e=np.abs(wrap(xf[:,2]-bearing))
hard=(it>40)|(st!=0)
print("hard frac",hard.mean())
for name,v in (("r",r),("|bearing-th0|",rel_b),("|yaw-th0|",rel_y),("|yaw-bearing|",rel_e)):
    qs=np.quantile(v,[0,.25,.5,.75,1]); 
    print(name,[f"{hard[(v>=qs[i])&(v<=qs[i+1])].mean():.2f}" for i in range(4)], "quartile edges",np.round(qs,2))
# reversing in solution?
rev=(uo[:,:-1,0]<-1e-3).any(1)
print("solution reverses: easy",rev[~hard&(st==0)].mean(),"hard",rev[hard&(st==0)].mean())
print("T easy",(do*49)[~hard].mean(),"T hard",(do*49)[hard&(st==0)].mean())
