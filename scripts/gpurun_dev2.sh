cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
PYTHONUNBUFFERED=1 timeout 500 python -u -m pytest tests -q -m gpu -k "dynamic_obstacles_with_turning or mixed_precision_meets or config2_full_batch_accounting or config5_candidates_vs_oracle" --durations=8 2>&1 | tail -14 | cut -c1-160
