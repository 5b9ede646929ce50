cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 120 python scripts/dev/dyn_hedge_repro.py two_circles 2>&1 | grep -v amdgpu.ids | tail -2
