#!/bin/bash
# per-kernel HIP-event averages of the bench's roofline leg under knob settings:  FILTER=sample tools/ab_knob_kernels.sh "" "du_tw=16"
for r in $(seq 1 ${ROUNDS:-2}); do
for cfg in "$@"; do
  K=""; for kv in $cfg; do K="$K --knob $kv"; done
  timeout -k 10 300 python bench.py --full-line $K --allow-nondefault-knobs --steps ${STEPS:-60} --warmup 3 --no-cpu-baseline --no-secondary ${EXTRA:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); ra=d['roofline_all']
print('%-24s img/s %7.1f digest %s | ' % ('[$cfg]', d['value'], d['score_digest']['match']) + '  '.join('%s %.1f' % (k.replace('k_','').replace('bottleneck','bnk'), v['avg_us']) for k,v in ra.items() if '${FILTER:-}' in k))" || exit $?
done; done
