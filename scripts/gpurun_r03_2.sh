cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
( time python -m pytest tests -m "gpu and not slow" -q -x --durations=5 ) > gpurun_out/r03/quick_tier_pit.log 2>&1
tail -15 gpurun_out/r03/quick_tier_pit.log
