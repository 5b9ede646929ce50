cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04; rm -rf gpurun_out/profile_summary
timeout 600 bash scripts/profile.sh r04_wave_kernel_caps100 carlike_n50_B1024_c4 > gpurun_out/r04/profile_a.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_caps100.json
BENCH_EXTRA="--caps 60,45,40,35" timeout 600 bash scripts/profile.sh r04_wave_kernel_caps60 carlike_n50_B1024_c4_caps60 > gpurun_out/r04/profile_b.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_caps60.json
BENCH_EXTRA="--batch 4096" timeout 600 bash scripts/profile.sh r04_wave_kernel_B4096 carlike_n50_B4096_c4 > gpurun_out/r04/profile_c.log 2>&1; cp gpurun_out/profile_summary/bench_under_rocprof.json gpurun_out/r04/bench_under_rocprof_B4096.json
grep -h "mpc_ipm" gpurun_out/profile_summary/*.md | cut -c1-120
timeout 600 python scripts/gpu_seed_parity_sweep.py 1 2 3 > gpurun_out/r04/seed_parity_sweep_c.log 2>&1; grep -E "^\[seed|^seed" gpurun_out/r04/seed_parity_sweep_c.log | cut -c1-200
timeout 400 python bench.py --force-dist --no-legs --no-cpu-baseline 2> gpurun_out/r04/bench_force_dist_c.err | grep '^{' > gpurun_out/r04/bench_force_dist_c.json; python -c "
import json; d=json.load(open('gpurun_out/r04/bench_force_dist_c.json')); print(d['value'], d['per_gpu_reference']['value'])"
for b in 256 1024 4096 16384 32768; do timeout 300 python bench.py --batch $b --caps 60,45,40,35 --no-legs --no-cpu-baseline --no-parity-check --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('batch', d['config'].get('batch_per_gpu'), 'value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'converged', d['solver']['converged_frac'])"; done > gpurun_out/r04/batch_sweep_caps60.log; cat gpurun_out/r04/batch_sweep_caps60.log
