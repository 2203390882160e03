#!/bin/bash
# One GPU-box session that regenerates the round's evidence under gpurun_out/ (copied into profiles/<round>_* by
# tools/collect_profiles.py afterwards): full-pool ENet bench line, ICNet bench line, rocprofv3 kernel stats and
# the PMC passes for both models.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python bench.py --detail-file $OUT/bench_enet_full_detail.json > $OUT/bench_enet_full.json 2> $OUT/bench_enet_full.log; rc=$?; echo "enet bench rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py --model icnet --steps 120 --warmup 3 --detail-file $OUT/bench_icnet_detail.json > $OUT/bench_icnet.json 2> $OUT/bench_icnet.log; rc=$?; echo "icnet bench rc=$rc"; [ $rc -ge 124 ] && exit $rc
for M in enet icnet; do
  # kernel stats of whole-batch launches on one stream (img_groups=1: the launch shape of the bench's roofline leg) ...
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final_$M -- python3 bench.py --model $M --steps 40 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --knob img_groups=1 --allow-nondefault-knobs > $OUT/rocprof_final_$M.log 2>&1; rc=$?; echo "rocprof $M rc=$rc"; [ $rc -ge 124 ] && exit $rc
  # ... and of the shipping schedule (two image-group chains: half-batch launches that overlap each other)
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final_${M}_groups2 -- python3 bench.py --model $M --steps 40 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/rocprof_final_${M}_groups2.log 2>&1; rc=$?; echo "rocprof $M groups2 rc=$rc"; [ $rc -ge 124 ] && exit $rc
  bash tools/gpu_pmc.sh $M || exit $?
done
# the opt-in bf16x3 kernels (never the headline): kernel stats + PMC of a run whose main leg scores in that mode
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final_enet_bf16x3 -- python3 bench.py --arithmetic bf16x3 --steps 40 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --knob img_groups=1 --allow-nondefault-knobs > $OUT/rocprof_final_enet_bf16x3.log 2>&1; rc=$?; echo "rocprof enet bf16x3 rc=$rc"; [ $rc -ge 124 ] && exit $rc
EXTRA="--arithmetic bf16x3" bash tools/gpu_pmc.sh enet enet_bf16x3 || exit $?
tail -c 400 $OUT/bench_enet_full.json; tail -c 400 $OUT/bench_icnet.json
