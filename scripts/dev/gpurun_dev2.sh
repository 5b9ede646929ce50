cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 200 python scripts/dev/perf_config3.py 2>&1 | grep -v amdgpu.ids
PYTHONUNBUFFERED=1 timeout 500 python -u -m pytest tests -q -m gpu -k "obstacle or config3 or clearance or footprint or rows or costmap or uninitialised or fleet or plugin" 2>&1 | tail -4 | cut -c1-200
