for b in 1 2 4 8 16; do
  steps=$((192 / b)); 
  python bench.py --batch $b --steps $steps --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch', $b, 'img/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'])"
done
