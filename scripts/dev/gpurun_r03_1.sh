set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time python -m pytest tests -m "gpu and not slow" -x -q --durations=15 ) > gpurun_out/r03/quick_tier.log 2>&1
tail -30 gpurun_out/r03/quick_tier.log
( time python -m pytest tests/test_gpu_stress.py -x -q -s ) > gpurun_out/r03/stress.log 2>&1
tail -8 gpurun_out/r03/stress.log
cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_f64_stage.hip -o /tmp/mfma_f64_stage 2>/dev/null && /tmp/mfma_f64_stage > $GRAFT_REPO_ROOT/gpurun_out/r03/mfma_f64_stage.log 2>&1; cd $GRAFT_REPO_ROOT
cat gpurun_out/r03/mfma_f64_stage.log
python bench.py --steps 20 --warmup 3 --no-legs --no-cpu-baseline > gpurun_out/r03/bench_base.json 2> gpurun_out/r03/bench_base.err
tail -c 1500 gpurun_out/r03/bench_base.json
