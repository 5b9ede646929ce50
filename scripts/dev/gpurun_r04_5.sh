cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r04/gpu_suite_10.log 2>&1; grep -E "passed|failed|^FAILED|^E   " gpurun_out/r04/gpu_suite_10.log | cut -c1-300 | tail -40
timeout 600 python bench.py > gpurun_out/r04/bench_10.json 2> gpurun_out/r04/bench_10.err; tail -c 6000 gpurun_out/r04/bench_10.json; tail -5 gpurun_out/r04/bench_10.err
