#!/bin/bash
(cd mpc_local_planner_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPC_POISON_LDS $1 mpc_capi.hip -o libmpc_hip.so)
python - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, mpc_local_planner_amd as m
for name, cfg in [("carlike_min_time_n50", m.config_carlike_min_time(50)), ("unicycle_quadratic_n20", m.config_unicycle_quadratic(20)), ("bicycle_min_time_n30", m.config_bicycle_min_time(30))]:
    g = np.load(f"tests/golden/{name}.npz")
    s = m.BatchSolver(cfg, max_batch=8)
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    print(name, "status", r.status.tolist(), "iters", r.iters.tolist(), "err", float(np.abs(r.x - g["x"]).max()))
PY
