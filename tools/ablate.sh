#!/bin/bash
# Phase / traffic ablation of the 128-channel bottleneck kernels (timing only, results invalid), same box, same build:
#   0 product behaviour   1 stop after the projection   2 no projection   3 no residual traffic   4 no store traffic
#   5 = 3 + 4   6 projection loads all hit pixel 0   7 = 5 + 6 (instruction stream only)
# needs libssal_hip_measure.so (python tools/phase_trace.py --build-measure); per-launch HIP-event averages of the bench's
# roofline leg.
LIBM=$(pwd)/semanticsegmentationactivelearning_amd/libssal_hip_measure.so
for ab in ${ABLATES:-0 1 2 3 4 5 6 7}; do
  SSAL_ABLATE=$ab SSAL_LIB_PATH=$LIBM timeout -k 10 300 python bench.py --full-line --allow-nondefault-knobs --allow-digest-mismatch --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['avg_launch_us_per_kernel'] if 'avg_launch_us_per_kernel' in d['roofline'] else d['roofline']['per_kernel_ms_per_batch']; print('ablate', $ab, {k: round(v,4) for k,v in r.items() if 'bottleneck_mfma' in k})" || exit $?
done
