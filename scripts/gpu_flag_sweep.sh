#!/bin/bash
# developer tool: compile the library with different hipcc flags and time the config-2 batch
cd mpc_local_planner_amd/csrc
for F in "-O3" "-O2" "-O3 -fno-unroll-loops" "-O3 -mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" "-O3 -mllvm -amdgpu-schedule-metric-bias=100" "-O3 -fno-slp-vectorize" "-Os"; do
  hipcc --offload-arch=gfx950 $F -std=c++17 -fPIC -shared mpc_capi.hip -o libmpc_hip.so 2>/dev/null || { echo "$F: build failed"; continue; }
  cd ../..
  echo -n "$F : "
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
  cd mpc_local_planner_amd/csrc
done
