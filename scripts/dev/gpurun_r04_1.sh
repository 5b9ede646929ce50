cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m "gpu and not slow" -q > gpurun_out/r04/quick_tier_1.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/r04/quick_tier_1.log | tail -40
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_1.json 2> gpurun_out/r04/bench_1.err; python -c "
import json;d=json.load(open('gpurun_out/r04/bench_1.json'));print({k:d[k] for k in ('value','ms_per_step')}); print(d.get('solver'))" | cut -c1-1500
