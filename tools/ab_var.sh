#!/bin/bash
# same-box A/B of library variants, interleaved ROUNDS times; an entry is lib.so or lib.so@"extra bench args":
#   LIBS='libssal_hip.so libssal_var_x.so@--knob_bnk_tw=16' FILTER=mfma tools/ab_var.sh     (underscores in args become spaces)
PKG=$(pwd)/semanticsegmentationactivelearning_amd
for r in $(seq 1 ${ROUNDS:-2}); do
for ent in $LIBS; do
  lib=${ent%%@*}; extra=""; [ "$ent" != "$lib" ] && extra=$(echo "${ent#*@}" | sed 's/--knob_/--knob /g')
  SSAL_LIB_PATH=$PKG/$lib timeout -k 10 300 python bench.py --full-line ${ARGS:-} $extra --allow-nondefault-knobs --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline --no-secondary --allow-digest-mismatch 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); ra=d['roofline_all']
print('%-40s img/s %7.1f digest %s | ' % ('$ent', d['value'], d['score_digest']['match']) + '  '.join('%s %.1f' % (k.replace('k_','').replace('bottleneck','bnk'), v['avg_us']) for k,v in ra.items() if '${FILTER:-}' in k))" || exit $?
done; done
