#!/bin/bash
# developer tool: per-iteration trace (+ census of non-finite LDS words) of instance $1 (default 3) of the car-like golden fixture;
# extra hipcc flags in $3 (e.g. -DMPC_POISON_LDS)
(cd mpc_local_planner_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $3 -DMPC_NANCHECK=${1:-3} mpc_capi.hip -o libmpc_hip.so)
python - <<'PY' 2>&1 | head -${2:-60}
import sys; sys.path.insert(0, '.')
import numpy as np, mpc_local_planner_amd as m
g = np.load("tests/golden/carlike_min_time_n50.npz")
s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=8)
r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
print("status", r.status.tolist(), "iters", r.iters.tolist())
PY
