cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_pitcheck.so timeout 300 python scripts/dev/pit_inertia_diag.py polygon > gpurun_out/r04/pitdiag_polygon.log 2>&1
grep -c "pit ok" gpurun_out/r04/pitdiag_polygon.log; grep -o "pit ok [01] (code [-0-9]*) serial ok [01]" gpurun_out/r04/pitdiag_polygon.log | sort | uniq -c
grep "code -1) serial ok 1" gpurun_out/r04/pitdiag_polygon.log | head -5 | cut -c1-200
