#!/bin/bash
# same-box A/B of the HIP runtime's hardware-queue cap (GPU_MAX_HW_QUEUES, default 4) against the image-group count:
# a process with more streams than hardware queues makes streams share a queue (ENet then ICNet in one process)
for r in $(seq 1 ${ROUNDS:-2}); do
for q in unset 2 4 8 16; do
for cfg in "img_groups=2" "img_groups=3" "img_groups=4"; do
  K=""; for kv in $cfg; do K="$K --knob $kv"; done
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 300 python bench.py $K --allow-nondefault-knobs --steps ${STEPS:-60} --warmup 3 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-10s %-16s img/s %7.1f  ms/step %.3f  digest %s' % ('hwq=$q', '[$cfg]', d['value'], d['ms_per_step'], d['score_digest']['match']))" || exit $?
done; done; done
# ENet followed by ICNet in one process (the secondary c4 leg): does the second model keep its speed?
for q in unset 8; do
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 400 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('%-10s enet %7.1f  c4 %7.1f  c5 %7.1f' % ('hwq=$q', d['value'], d['secondary']['c4']['value'], d['secondary']['c5']['value']))" || exit $?
done
