cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python scripts/gpu_seed_parity_sweep.py 1 2 3 > gpurun_out/r04/seed_parity_sweep_b.log 2>&1; grep -E "^\[seed|^seed" gpurun_out/r04/seed_parity_sweep_b.log | cut -c1-330
timeout 300 python __graft_entry__.py smoke > gpurun_out/r04/smoke_b.log 2>&1; tail -3 gpurun_out/r04/smoke_b.log
timeout 400 python bench.py --force-dist --no-legs --no-cpu-baseline > gpurun_out/r04/bench_force_dist_b.json 2> gpurun_out/r04/bench_force_dist_b.err; python -c "
import json; d=json.load(open('gpurun_out/r04/bench_force_dist_b.json')); print(d['value'], d['n_gpus'], d.get('config'), d.get('per_gpu_reference'))"
