cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
PYTHONUNBUFFERED=1 timeout 300 python -u -m pytest tests/test_gpu_reference_plugin.py -q -s -m gpu > gpurun_out/r03/plugin_tests.log 2>&1; grep "SUCCESS\|passed\|failed" gpurun_out/r03/plugin_tests.log | cut -c1-230
