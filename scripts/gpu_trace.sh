#!/bin/bash
# debug: trace instance 0 of the unicycle golden batch on the GPU (kernel printf)
set -e
cd mpc_local_planner_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared '-DMPC_TRACE' '-DMPC_TRACE_COND=(blockIdx.x==0&&threadIdx.x==0)' mpc_capi.hip -o libmpc_hip.so
cd ../..
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.')
import mpc_local_planner_amd as m
g = np.load("tests/golden/unicycle_quadratic_n20.npz")
s = m.BatchSolver(m.config_unicycle_quadratic(20, max_iter=40), max_batch=8)
r = s.solve(g["x0"][:1], g["xf"][:1], g["u_prev"][:1], g["dt_prev"][:1])
print(r.status, r.iters)
PY
