cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r04/bench_6.json 2> gpurun_out/r04/bench_6.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_6.json'))
print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['solver'].get('answers_equal_to_the_reference_path_alone'), d['solver']['converged_frac'], d['solver']['iters_mean'])
for k,v in d['legs'].items():
    if 'value' in v: print("   ",k, round(v['value']), round(v.get('ms_per_step',0),3), v.get('solver',{}).get('iters_mean', v.get('iters_mean')), v.get('solver',{}).get('converged_frac', v.get('converged_frac')))
    else: print("   ",k,v.get('ms_p50'))
PY
timeout 600 python -m pytest tests -m "gpu and not slow" -q -x 2>&1 | tail -3
