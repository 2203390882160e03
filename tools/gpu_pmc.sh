#!/bin/bash
# PMC passes (separate runs, --kernel-trace only next to --pmc): SQ pipe counters, then HBM bytes.
# Usage: tools/gpu_pmc.sh [enet|icnet] [tag]      EXTRA="--arithmetic bf16x3" tools/gpu_pmc.sh enet enet_bf16x3
set -u
MODEL=${1:-enet}
TAG=${2:-$MODEL}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# img_groups=1: whole-batch launches on one stream, the launch shape bench.py's roofline leg times (ssal_profile_enable
# serialises the image groups); per-launch traffic of the default two-chain schedule would be that of half-batch launches
CMD="python3 bench.py --model $MODEL ${EXTRA:-} --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --knob img_groups=1 --allow-nondefault-knobs"
run() { # name counters...
    local name=$1; shift
    echo "=== pmc $TAG $name: $*" | tee -a $OUT/pmc_summary.log
    timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$name -- $CMD > $OUT/pmc_${TAG}_$name.log 2>&1
    local rc=$?
    echo "pmc $TAG $name rc=$rc" | tee -a $OUT/pmc_summary.log
    if [ $rc -ge 124 ]; then exit $rc; fi
}
: > $OUT/pmc_summary.log
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
echo "=== done" | tee -a $OUT/pmc_summary.log
