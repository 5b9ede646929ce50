cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
MPC_HIP_LIB=$GRAFT_REPO_ROOT/mpc_local_planner_amd/csrc/libmpc_hip_pitcheck.so timeout 300 python scripts/dev/pit_inertia_diag.py > gpurun_out/r04/pitdiag_headline.log 2>&1
grep -c "pit ok" gpurun_out/r04/pitdiag_headline.log; grep -o "pit ok [01] (code [-0-9]*) serial ok [01]" gpurun_out/r04/pitdiag_headline.log | sort | uniq -c; tail -1 gpurun_out/r04/pitdiag_headline.log
timeout 300 python bench.py --no-legs --no-cpu-baseline > gpurun_out/r04/bench_15.json 2> gpurun_out/r04/bench_15.err; python -c "
import json; d=json.load(open('gpurun_out/r04/bench_15.json')); print(d['value'], d['ms_per_step'], d['solver']['converged_frac'], d['solver']['answers_equal_to_the_reference_path_alone'])"
