#!/bin/bash
# developer tool: build with per-phase cycle counters, run one batch, print where the slowest waves spend their time
set -e
cd mpc_local_planner_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPC_PROFILE=1 mpc_capi.hip -o libmpc_hip.so
cd ../..
python - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _lib
B, n = 1024, 50
s = m.BatchSolver(m.config_carlike_min_time(n), max_batch=B)
x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B)
r = s.solve(x0, xf, up, dtp)
r = s.solve(x0, xf, up, dtp)
print("kernel ms", s.last_kernel_ms())
lib = _lib.load()
buf = np.zeros((B, 16), dtype=np.int64)
lib.mpc_debug_profile(buf.ctypes.data_as(C.c_void_p), C.c_int(B))
names = ["ticks", "wall100MHz", "iters", "nfac", "ntrial", "kkt", "barrier_terms", "backward", "forward", "post", "logs0", "trial", "accept", "bwd_loop", "bwd_setup", "fwd_loop"]
o = np.argsort(-buf[:, 0])[:5]
print("tick rate GHz ~", (buf[:, 0] / (buf[:, 1] / 100e6)).mean() / 1e9)
for i in o:
    print(i, dict(zip(names, buf[i, :16].tolist())))
print("mean over waves:", dict(zip(names, buf[:, :16].mean(0).round(0).tolist())))
print("per-sweep ticks: backward", buf[:, 7].sum() / buf[:, 3].sum(), "forward", buf[:, 8].sum() / buf[:, 3].sum(), "trial", buf[:, 11].sum() / max(1, buf[:, 4].sum()))
PY
