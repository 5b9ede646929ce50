cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_9.json 2> gpurun_out/r04/bench_9.err
bash scripts/profile.sh r04_wave_kernel_caps100 carlike_n50_B1024_c4 > gpurun_out/r04/profile_caps100.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_caps100.json
BENCH_EXTRA="--caps 60,45,40,35" bash scripts/profile.sh r04_wave_kernel_caps60 carlike_n50_B1024_c4_caps60 > gpurun_out/r04/profile_caps60.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_caps60.json
BENCH_EXTRA="--batch 4096" bash scripts/profile.sh r04_wave_kernel_B4096 carlike_n50_B4096_c4 > gpurun_out/r04/profile_B4096.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_B4096.json
python scripts/gpu_batch_sweep.py > gpurun_out/r04/batch_sweep_caps60.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 --force-dist --no-cpu-baseline > gpurun_out/r04/bench_force_dist.json 2> gpurun_out/r04/bench_force_dist.err; tail -c 1200 gpurun_out/r04/bench_force_dist.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_9.json'))
print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['solver'].get('answers_equal_to_the_reference_path_alone'), d['solver']['converged_frac'], d['solver']['iters_mean'])
for k,v in d['legs'].items():
    if 'value' in v: print("   ",k, round(v['value']), round(v.get('ms_per_step',0),3), v.get('solver',{}).get('iters_mean', v.get('iters_mean')), v.get('solver',{}).get('converged_frac', v.get('converged_frac')))
    else: print("   ",k,v.get('ms_p50'))
PY
