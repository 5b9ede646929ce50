cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
( time python -m pytest tests -m "gpu" -q -s --durations=8 ) > gpurun_out/r03/full_gpu_suite.log 2>&1
grep -v "^HIP\|^ROCm\|^Host\|^Librccl\|^RCCL\|amdgpu.ids" gpurun_out/r03/full_gpu_suite.log | tail -40 | cut -c1-300
