cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
export MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_pitobst.so
for mu in 1e-8 1e-7 1e-6 1e-5 1e-4; do MPC_PIT_MU=$mu timeout 200 python scripts/dev/pit_headline_sweep.py 2>&1 | tail -1; done > gpurun_out/r04/pit_headline_sweep.log
MPC_NO_PIT=1 timeout 200 python scripts/dev/pit_headline_sweep.py 2>&1 | tail -1 >> gpurun_out/r04/pit_headline_sweep.log
cat gpurun_out/r04/pit_headline_sweep.log
