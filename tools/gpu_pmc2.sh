#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline"
run() { local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc2_$name -- $CMD > $OUT/pmc2_$name.log 2>&1
    local rc=$?; echo "pmc2 $name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi; }
run sq2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES
run sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES
run sq4 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES SQ_CYCLES
run ta TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo done
