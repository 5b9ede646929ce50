cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 300 python -m pytest tests/test_gpu_reference_plugin.py -x -q -s -m gpu --durations=15 > gpurun_out/r03/plugin_tests.log 2>&1; echo rc=$?; tail -25 gpurun_out/r03/plugin_tests.log | cut -c1-250
