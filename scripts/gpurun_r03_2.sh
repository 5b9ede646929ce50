cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 300 python scripts/dev/pit_check.py ab 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/pit_ab2.log
export MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_prof.so
echo "== pit profile (all PIT)"; MPC_PIT_MU=0 timeout 200 python scripts/gpu_phase_profile.py 2>&1 | grep "per iteration\|per-sweep\|kernel ms"
