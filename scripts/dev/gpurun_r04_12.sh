cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_prof.so timeout 300 python scripts/dev/phase_profile.py > gpurun_out/r04/phase_profile_b.log 2>&1; tail -6 gpurun_out/r04/phase_profile_b.log | cut -c1-600
