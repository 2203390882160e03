#!/bin/bash
# same-box A/B over launcher knobs of the PRODUCT library (bench.py --knob):  tools/ab_knobs.sh "" "img_groups=2" "img_groups=4"
for r in $(seq 1 ${ROUNDS:-2}); do
for cfg in "$@"; do
  K=""; for kv in $cfg; do K="$K --knob $kv"; done
  timeout -k 10 300 python bench.py $K --allow-nondefault-knobs --steps ${STEPS:-60} --warmup 3 --no-cpu-baseline --no-secondary --no-roofline ${EXTRA:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-28s img/s %7.1f  ms/step %.3f  digest %s' % ('[$cfg]', d['value'], d['ms_per_step'], d['score_digest']['match']))" || exit $?
done; done
