#!/bin/bash
# rocprofv3 runs for profiles/: kernel trace + stats, then PMC counters in separate passes.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o run -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o run -- $CMD > $OUT/pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -50
tail -2 $OUT/bench_stats.log
