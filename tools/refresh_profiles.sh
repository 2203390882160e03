#!/bin/bash
# One GPU-box session that regenerates everything under profiles/r01_final_* and profiles/r01_pmc:
# full-pool bench line, rocprofv3 kernel stats, the PMC passes, the memory-pattern / MFMA probes.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.log; rc=$?; echo "bench rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/rocprof_final.log 2>&1; rc=$?; echo "rocprof rc=$rc"; [ $rc -ge 124 ] && exit $rc
bash tools/gpu_pmc.sh || exit $?
timeout -k 10 120 python tools/mem_probe.py 8 > $OUT/mem_probe.log 2>&1; echo "mem_probe rc=$?"
timeout -k 10 120 python tools/hbm_bw.py > $OUT/hbm_bw.log 2>&1; echo "hbm_bw rc=$?"
timeout -k 10 200 python tools/mfma_peak.py > $OUT/mfma_peak.log 2>&1; echo "mfma_peak rc=$?"
tail -c 600 $OUT/bench_full.json
