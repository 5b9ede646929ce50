cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
BENCH_EXTRA="--batch 4096" bash scripts/profile.sh r03_wave_kernel_B4096 carlike_n50_B4096_c4 > gpurun_out/profile_run_B4096.log 2>&1
grep -n "mpc_ipm\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST_VALU\|SQ_WAIT_ANY\|GRBM\|SQ_BUSY" gpurun_out/profile_summary/r03_wave_kernel_B4096.md | cut -c1-150
cat gpurun_out/profile_summary/bench_under_rocprof.json | cut -c1-300
