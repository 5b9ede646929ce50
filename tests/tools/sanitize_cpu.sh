#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer on the CPU builds (GPU sanitizers are not available on the pool): the C oracle on the three BASELINE shapes and the host build
# of the kernel core (csrc/mpc_core.hpp through tests/host_harness/host_solver.cpp).  usage: bash tests/tools/sanitize_cpu.sh   (r06: both clean)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -fPIC -shared $R/oracle/mpc_oracle.c -o $T/liboracle_asan.so -lm
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared $R/tests/host_harness/host_solver.cpp -o $T/libhost_asan.so
cat > $T/run.py <<PY
import sys, ctypes as C
sys.path.insert(0, "$R")
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import _abi as A, workloads as W
CO.LIB = "$T/liboracle_asan.so"; CO._lib = None
print("C oracle, car-like n = 50:", (CO.solve_batch(CO.from_nlp_config(R.config_carlike_min_time(50)), *W.carlike_min_time_inputs(64), nthreads=2)[3] == 0).mean())
x0, xf, up, dtp, obs = W.unicycle_obstacle_inputs(32, n_obst=16, max_vertices=6, lateral=(0.15, 0.8))
c3 = R.config_unicycle_quadratic(80)
print("C oracle, config 3:", (CO.solve_batch(CO.from_nlp_config(c3), x0, xf, up, dtp, obstacles=obs, obst=CO.obst_from_nlp_config(c3, 16, 6, 4), nthreads=2)[3] == 0).mean())
print("C oracle, bicycle n = 120:", (CO.solve_batch(CO.from_nlp_config(R.config_bicycle_min_time(120)), *W.bicycle_min_time_inputs(16), nthreads=2)[3] == 0).mean())
lib = C.CDLL("$T/libhost_asan.so")
for name, cfg, inp in (("car-like n = 20", A.config_carlike_min_time(20), W.carlike_min_time_inputs(8, seed=3, goal_range=(1.0, 2.5))), ("unicycle n = 20", A.config_unicycle_quadratic(20), W.unicycle_quadratic_inputs(8, seed=4)),
                       ("bicycle n = 30", A.config_bicycle_min_time(30), W.carlike_min_time_inputs(8, seed=5, goal_range=(2.0, 6.0)))):
    x0, xf, up, dtp = inp; B, n = 8, cfg.n
    xo = np.zeros((B, n, 3)); uo = np.zeros((B, n, 2)); do = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros(B)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.hostdbg_solve(C.byref(cfg), C.c_int(B), p(x0), p(xf), p(up), p(dtp), None, None, None, p(xo), p(uo), p(do), p(st), p(it), p(kkt))
    print("kernel core on the host,", name, "status", st.tolist())
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python $T/run.py
rm -rf $T
echo "sanitizers: clean"
