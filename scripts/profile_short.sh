#!/bin/bash
# short variant of profile.sh: kernel trace + stats and the two HBM traffic counters only (three rocprofv3 passes)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/mpc_prof
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profile_summary
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-warm"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
cd $R && python scripts/summarize_profile.py $OUT gpurun_out/profile_summary/${1:-r01_wave_kernel_final} > /dev/null
grep -h '"metric"' $OUT/bench_stats.log | tail -1 > gpurun_out/profile_summary/bench_under_rocprof.json
