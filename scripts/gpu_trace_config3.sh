#!/bin/bash
# developer tool: per-iteration trace of instance $1 (default 6) of the config-3-shaped batch (unicycle quadratic, n=80, 16 polygons)
(cd mpc_local_planner_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPC_NANCHECK=${1:-6} mpc_capi.hip -o libmpc_hip.so)
python - <<'PY' 2>&1 | grep "ls:\|ls FAILED\|status" | head -${2:-40} | cut -c1-330
import sys; sys.path.insert(0, '.')
import numpy as np, mpc_local_planner_amd as m
n, B, O, V, M = 80, 256, 16, 6, 4
x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), max_batch=B)
r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
print("status", r.status[:12].tolist(), "iters", r.iters[:12].tolist())
PY
