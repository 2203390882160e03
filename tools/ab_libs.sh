#!/bin/bash
# same-box A/B of library variants (tools/build_variant.sh): per-launch HIP-event averages of the bench's roofline leg
# and images/s, interleaved ROUNDS times.   tools/ab_libs.sh libA.so libB.so ...     (paths relative to the package dir)
PKG=$(pwd)/semanticsegmentationactivelearning_amd
for r in $(seq 1 ${ROUNDS:-2}); do
for lib in "$@"; do
  SSAL_LIB_PATH=$PKG/$lib timeout -k 10 300 python bench.py --full-line --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline --no-secondary --allow-digest-mismatch 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); ra=d['roofline_all']
print('%-28s img/s %7.1f  digest %s | ' % ('$lib', d['value'], d['score_digest']['match']) + '  '.join('%s %.1f' % (k.replace('k_',''), v['avg_us']) for k,v in ra.items() if '${FILTER:-mfma}' in k))" || exit $?
done; done
