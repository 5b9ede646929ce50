# round-3 closing run: smoke() and the whole GPU suite with the accounting lines (unbuffered log)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 300 python __graft_entry__.py smoke > gpurun_out/r03/smoke.log 2>&1; echo smoke rc=$?; grep -v amdgpu.ids gpurun_out/r03/smoke.log | tail -6 | cut -c1-250
PYTHONUNBUFFERED=1 timeout 1500 python -u -m pytest tests -q -m gpu -s --durations=12 > gpurun_out/r03/full_gpu_suite_final.log 2>&1; echo suite rc=$?; tail -18 gpurun_out/r03/full_gpu_suite_final.log | cut -c1-200
