cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 100 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lds_working_set or fixed_layout" 2>&1 | tail -2
timeout 200 python bench.py --steps 20 --warmup 3 > gpurun_out/r03/bench_final2.json 2> gpurun_out/r03/bench_final2.err; python -c "
import json; d=json.loads(open('gpurun_out/r03/bench_final2.json').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['config']['lds_bytes_per_instance'], d['config']['workgroups_per_cu'], {k:(v.get('workgroups_per_cu'), round(v.get('value',0))) for k,v in d['legs'].items()})"
