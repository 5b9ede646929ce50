cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_clearance.py tests/test_gpu_parity.py -m gpu -q -s -k "clearance or terminal_ball or non_finite or hedged_answers" > gpurun_out/r04/gpu_part_4.log 2>&1; grep -E "passed|failed|^FAILED|clearance to every|^E  " gpurun_out/r04/gpu_part_4.log | cut -c1-300 | tail -20
