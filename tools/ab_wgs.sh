#!/bin/bash
# A/B of the workgroups-per-CU budget of k_bottleneck_mfma (same build, same box)
for w in 2 3; do
  SSAL_BNK_WGS=$w python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['per_kernel_ms_per_batch']; print('wgs', $w, 'img/s', d['value'], {k: round(v,3) for k,v in r.items()})"
done
