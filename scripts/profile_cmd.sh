#!/bin/bash
# rocprofv3 around one workload of scripts/gpu_solve_once.py: kernel trace + stats, then PMC counters in separate passes (never combined with sys/hip traces).
#   scripts/profile_cmd.sh <summary name under gpurun_out/profile_summary/> <key of the record in hbm_traffic.json / pmc_counters.json> <workload> [launches]
R=$GRAFT_REPO_ROOT
NAME=$1; KEY=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/mpc_prof_$NAME
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profile_summary
CMD="python $R/scripts/gpu_solve_once.py $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/run.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o run -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o run -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_LDS -d $OUT/pmc_vmem -o run -- $CMD > $OUT/pmc_vmem.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr -d $OUT/pmc_cache -o run -- $CMD > $OUT/pmc_cache.log 2>&1
cd $R && python scripts/summarize_profile.py $OUT gpurun_out/profile_summary/$NAME $KEY > /dev/null
grep -h "stage_data=" $OUT/run.log | tail -1 >> gpurun_out/profile_summary/$NAME.md
