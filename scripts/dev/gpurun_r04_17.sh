cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 500 python scripts/gpu_candidate_sweep_config5.py 1024 2>/dev/null | grep '^{' > gpurun_out/r04/candidate_sweep_config5.log
python - <<'PY'
import json
for l in open('gpurun_out/r04/candidate_sweep_config5.log'):
    d = json.loads(l); print(d['seed'], d['kinds'], d['caps'], d['kernel_ms'], d['converged'], d['iters_total'], d['conv_solves_per_s'], d['winners'])
PY
