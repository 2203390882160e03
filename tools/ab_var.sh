#!/bin/bash
# same-box A/B of library variants with extra bench args:  LIBS="libssal_hip.so libssal_var_x.so" ARGS="--knob bnk_tw=16" tools/ab_var.sh
PKG=$(pwd)/semanticsegmentationactivelearning_amd
for r in $(seq 1 ${ROUNDS:-2}); do
for lib in $LIBS; do
  SSAL_LIB_PATH=$PKG/$lib timeout -k 10 300 python bench.py --full-line ${ARGS:-} --allow-nondefault-knobs --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline --no-secondary --allow-digest-mismatch 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); ra=d['roofline_all']
print('%-24s img/s %7.1f digest %s | ' % ('$lib', d['value'], d['score_digest']['match']) + '  '.join('%s %.1f' % (k.replace('k_','').replace('bottleneck','bnk'), v['avg_us']) for k,v in ra.items() if '${FILTER:-}' in k))" || exit $?
done; done
