cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r04/gpu_suite_11.log 2>&1; grep -E "passed|failed|^FAILED|^E   |second-order check" gpurun_out/r04/gpu_suite_11.log | cut -c1-400 | tail -20
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 bash scripts/profile.sh r04b > gpurun_out/r04/profile_r04b.log 2>&1; tail -5 gpurun_out/r04/profile_r04b.log
ls gpurun_out | head; ls profiles | tail -5
