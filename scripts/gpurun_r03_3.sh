cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
( time python -m pytest tests -m "gpu and slow" -q --durations=8 ) > gpurun_out/r03/slow_tier_pit.log 2>&1
tail -30 gpurun_out/r03/slow_tier_pit.log
