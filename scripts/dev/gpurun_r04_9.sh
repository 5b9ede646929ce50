cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04; rm -rf gpurun_out/profile_summary
timeout 600 bash scripts/profile.sh r04_wave_kernel_caps100 carlike_n50_B1024_c4 > gpurun_out/r04/profile_a.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_caps100.json
BENCH_EXTRA="--caps 60,45,40,35" timeout 600 bash scripts/profile.sh r04_wave_kernel_caps60 carlike_n50_B1024_c4_caps60 > gpurun_out/r04/profile_b.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_caps60.json
BENCH_EXTRA="--batch 4096" timeout 600 bash scripts/profile.sh r04_wave_kernel_B4096 carlike_n50_B4096_c4 > gpurun_out/r04/profile_c.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_B4096.json
ls gpurun_out/profile_summary; grep -h "mpc_ipm" gpurun_out/profile_summary/*.md | cut -c1-120
