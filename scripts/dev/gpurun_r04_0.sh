cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests -m "gpu and not slow" -x -q > gpurun_out/r04/quick_tier_0.log 2>&1; tail -3 gpurun_out/r04/quick_tier_0.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_0.json 2> gpurun_out/r04/bench_0.err; tail -c 1500 gpurun_out/r04/bench_0.json
