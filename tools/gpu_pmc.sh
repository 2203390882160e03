#!/bin/bash
# PMC passes (separate runs, --kernel-trace only next to --pmc): SQ pipe counters, then HBM bytes.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
run() { # name timeout counters...
    local name=$1; shift
    echo "=== pmc $name: $*" | tee -a $OUT/pmc_summary.log
    timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1
    local rc=$?
    echo "pmc $name rc=$rc" | tee -a $OUT/pmc_summary.log
    if [ $rc -ge 124 ]; then exit $rc; fi
}
: > $OUT/pmc_summary.log
if [ "${LIST_COUNTERS:-0}" = "1" ]; then timeout 120 rocprofv3 -L > $OUT/counters.txt 2>&1; fi
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
echo "=== done" | tee -a $OUT/pmc_summary.log
ls -R $OUT | grep -c counter_collection
