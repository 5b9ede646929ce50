cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 300 python scripts/dev/occupancy_probe.py 2>&1 | tail -1 > gpurun_out/r04/occupancy_probe.log
MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_w2.so timeout 300 python scripts/dev/occupancy_probe.py 2>&1 | tail -1 >> gpurun_out/r04/occupancy_probe.log
cat gpurun_out/r04/occupancy_probe.log
