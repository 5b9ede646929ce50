#!/bin/bash
# developer tool: swap the generated DPP asm blocks for single-instruction helpers (bit mask: 1 T1, 2 H, 4 V, 8 FWD) and check a golden fixture
for m in "$@"; do
(cd mpc_local_planner_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPC_DPP_DEBUG=$m mpc_capi.hip -o libmpc_hip.so)
python - <<PY
import sys; sys.path.insert(0, '.')
import numpy as np, mpc_local_planner_amd as m
g = np.load("tests/golden/carlike_min_time_n50.npz")
s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=8)
r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
print("mask $m status", r.status.tolist(), "iters", r.iters.tolist(), "golden", g["iters"].tolist(), "err", float(np.abs(r.x - g["x"]).max()))
PY
done
