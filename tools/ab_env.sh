#!/bin/bash
# A/B of launcher knobs (same build, same box): each argument is an env assignment list, e.g.
#   tools/ab_env.sh "SSAL_BNK_XCD=0" "SSAL_BNK_XCD=1 SSAL_BNK_TW=16"      (BATCH=16 STEPS=12 to change the step)
for cfg in "$@"; do
  env $cfg SSAL_LIB_PATH=$(pwd)/semanticsegmentationactivelearning_amd/libssal_hip_trace.so python bench.py --full-line --allow-nondefault-knobs --batch ${BATCH:-8} --steps ${STEPS:-16} --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['per_kernel_ms_per_batch']; print('$cfg', 'img/s %.1f' % d['value'], {k: round(v,3) for k,v in r.items()})"
done
