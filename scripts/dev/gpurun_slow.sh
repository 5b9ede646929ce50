cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
nproc; lscpu | grep "Model name\|^CPU(s)"
PYTHONUNBUFFERED=1 timeout 1700 python -u -m pytest tests -q -m "gpu and slow" -s --durations=25 > gpurun_out/r03/slow_tier.log 2>&1; echo rc=$?; tail -32 gpurun_out/r03/slow_tier.log | cut -c1-200
