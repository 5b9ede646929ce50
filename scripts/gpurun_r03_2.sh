cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 300 python scripts/dev/obst_candidates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/obst_candidates.log
python -m pytest tests/test_gpu_parity.py -q -x -k "line_footprint_golden or fp32_path or active_clearance or obstacle_rows_golden" -s 2>&1 | tail -8 | cut -c1-250
python bench.py --steps 20 --warmup 3 > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err; tail -c 600 gpurun_out/r03/bench_full.err; python -c "
import json; d=json.load(open('gpurun_out/r03/bench_full.json')); print(d['value'], d['ms_per_step'], d['solver']); [print(k, {a: b for a, b in v.items() if a in ('value','ms_per_step','instances_with_an_active_clearance_row','answers_equal_to_the_reference_path_alone','ms_p50')}, v.get('solver',{}).get('converged_frac')) for k, v in d['legs'].items()]; print(d.get('cpu_baseline',{}).get('value'))"
