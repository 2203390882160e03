#!/bin/bash
# throughput at the frame sizes the reference's own parameter files configure (conf/*.json:33-34), batch 8
run() {
  python bench.py --full-line "$@" --steps 96 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']['per_kernel_ms_per_batch']
print('$*', '-> img/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'sum of kernel ms %.3f' % sum(r.values()))"
}
run --height 512 --width 1024
run --height 432 --width 648 --channels 4 --classes 6
run --height 256 --width 512 --batch 4
