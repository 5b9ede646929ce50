cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r04/gpu_suite_17.log 2>&1; grep -E "passed|failed|^FAILED|^E   |second-order check" gpurun_out/r04/gpu_suite_12.log | cut -c1-300 | tail -20
timeout 600 python bench.py > gpurun_out/r04/bench_17.json 2> gpurun_out/r04/bench_17.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/bench_17.json'))
print(d['value'], d['ms_per_step'], d['solver']['answers_equal_to_the_reference_path_alone'], d['solver']['converged_frac'], d['solver']['reference_path_alone_converged_frac'])
for k, v in d.get('legs', {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('solver', {}).get('converged_frac'), v.get('solver', {}).get('iters_mean'), v.get('ms_mean'))
PY
tail -3 gpurun_out/r04/bench_17.err
