cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_abi.py tests/test_gpu_closed_loop.py -m gpu -q -s -k "time_limit or golden or determin or closed" > gpurun_out/r04/gpu_part_14.log 2>&1; grep -E "passed|failed|^FAILED|^E   |time limit" gpurun_out/r04/gpu_part_14.log | cut -c1-300 | tail
