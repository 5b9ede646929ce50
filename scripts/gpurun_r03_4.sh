# round-3 measurement run: full GPU suite (with the accounting lines), bench line, rocprof summary, batch sweep, phase timers (partitioned vs serial sweeps)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -q -m gpu -s --durations=8 > gpurun_out/r03/full_gpu_suite.log 2>&1; tail -3 gpurun_out/r03/full_gpu_suite.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r03/bench_full3.json 2> gpurun_out/r03/bench_full3.err
bash scripts/profile.sh r03_wave_kernel_pit_v4 > gpurun_out/profile_run.log 2>&1
python scripts/gpu_batch_sweep.py > gpurun_out/r03/batch_sweep.log 2>&1
(echo "== serial sweeps (MPC_NO_PIT=1)"; MPC_NO_PIT=1 MPC_HIP_LIB=$PWD/mpc_local_planner_amd/csrc/libmpc_hip_prof.so python scripts/gpu_phase_profile.py; echo "== partitioned sweeps"; MPC_HIP_LIB=$PWD/mpc_local_planner_amd/csrc/libmpc_hip_prof.so python scripts/gpu_phase_profile.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/phase_profile_final.log
tail -12 gpurun_out/r03/batch_sweep.log
