cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r04/gpu_suite_final.log 2>&1; grep -E "passed|failed|^FAILED|^E   " gpurun_out/r04/gpu_suite_final.log | cut -c1-300 | tail -8
timeout 300 python __graft_entry__.py smoke > gpurun_out/r04/smoke_final.log 2>&1; tail -3 gpurun_out/r04/smoke_final.log
timeout 600 python bench.py --no-legs > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/bench_final.json') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['solver']['answers_equal_to_the_reference_path_alone'], d['solver']['converged_frac'], d['roofline']['frac'], d['cpu_baseline']['value'])"
