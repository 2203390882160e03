#!/bin/bash
# throughput vs batch size
for b in ${BATCHES:-1 2 4 6 8 12 16 24}; do
  steps=$((192 / b));
  python bench.py --full-line --batch $b --steps $steps --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['per_kernel_ms_per_batch']; print('batch', $b, 'img/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'bnk_mfma us/img %.2f' % (1000*(r['k_bottleneck_mfma<32>']+r.get('k_bottleneck_mfma<16>',0))/$b), 'b16 us/img %.2f' % (1000*r['k_bottleneck16<32,64,16>']/$b))"
done
