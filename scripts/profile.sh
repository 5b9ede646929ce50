#!/bin/bash
# rocprofv3 runs for profiles/: kernel trace + stats, then PMC counters in separate passes (never combined with sys/hip traces).
# The result databases stay on the GPU box (/tmp); only the text summaries come back through gpurun_out/.
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/mpc_prof
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profile_summary
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --no-parity-check ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o run -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o run -- $CMD > $OUT/pmc_lds.log 2>&1
cd $R && python scripts/summarize_profile.py $OUT gpurun_out/profile_summary/${1:-r02_wave_kernel} ${2:-carlike_n50_B1024_c4} > /dev/null
grep -h '"metric"' $OUT/bench_stats.log | tail -1 > gpurun_out/profile_summary/bench_under_rocprof.json
ls -la gpurun_out/profile_summary
