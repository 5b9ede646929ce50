cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
python scripts/dev/perf_quick.py > gpurun_out/r03/perf_quick.log 2>&1
grep -v amdgpu.ids gpurun_out/r03/perf_quick.log
timeout 600 python -m pytest tests -x -q -m "gpu and not slow" > gpurun_out/r03/quick_tier.log 2>&1; tail -3 gpurun_out/r03/quick_tier.log | grep "passed\|failed"
