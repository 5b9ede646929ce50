cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
export MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_dev.so
for v in "" "MPC_NO_PIT=1" "MPC_PIT_MU=1e-2" "MPC_PIT_MU=1e-3"; do echo "== $v"; env $v timeout 400 python -m pytest tests/test_gpu_closed_loop.py -m gpu -q -s -k "config5_candidates" 2>&1 | grep -E "config 5 candidates|config 5 shape with candidates|passed|failed" | cut -c1-250; done > gpurun_out/r04/mixed_pit_probe.log
cat gpurun_out/r04/mixed_pit_probe.log
