cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
python bench.py --steps 20 --warmup 3 > gpurun_out/r03/bench_full2.json 2> gpurun_out/r03/bench_full2.err
bash scripts/profile.sh r03_wave_kernel_pit_v3 > gpurun_out/profile_run.log 2>&1
python scripts/gpu_batch_sweep.py > gpurun_out/r03/batch_sweep.log 2>&1
tail -12 gpurun_out/r03/batch_sweep.log
