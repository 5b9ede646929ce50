#!/bin/bash
# developer tool: hunt the intermittent LINSOLVE failure (fresh process each time)
for variant in "" "-DMPC_UNIFORM_SWEEP"; do
  (cd mpc_local_planner_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $variant mpc_capi.hip -o libmpc_hip.so)
  echo "variant: [$variant]"
  for i in 1 2 3 4 5 6; do
    python - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, mpc_local_planner_amd as m
g = np.load("tests/golden/carlike_min_time_n50.npz")
s = m.BatchSolver(m.config_carlike_min_time(50), max_batch=8)
r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
r2 = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
print("first", r.status.tolist(), r.iters.tolist(), "second", r2.status.tolist())
PY
  done
done
