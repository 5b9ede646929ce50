cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
( time python -m pytest tests -m "gpu and not slow" -q -x ) > gpurun_out/r03/quick_tier_v4.log 2>&1
grep "passed\|failed" gpurun_out/r03/quick_tier_v4.log | tail -2
timeout 300 python scripts/dev/pit_check.py ab 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/pit_ab4.log | cut -c1-130
