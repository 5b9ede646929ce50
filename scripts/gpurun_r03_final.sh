# round-3 closing run: smoke(), the whole GPU suite with the accounting lines (unbuffered log), the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 300 python __graft_entry__.py smoke > gpurun_out/r03/smoke.log 2>&1; echo smoke rc=$?; grep -v amdgpu.ids gpurun_out/r03/smoke.log | tail -4 | cut -c1-250
PYTHONUNBUFFERED=1 timeout 900 python -u -m pytest tests -q -m gpu -s --durations=8 > gpurun_out/r03/full_gpu_suite_final.log 2>&1; echo suite rc=$?; tail -12 gpurun_out/r03/full_gpu_suite_final.log | cut -c1-200
python bench.py --steps 20 --warmup 3 > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err; tail -c 300 gpurun_out/r03/bench_final.json
