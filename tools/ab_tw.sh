for cfg in "" "--knob bnk_tw=16"; do
python bench.py --full-line $cfg --allow-nondefault-knobs --steps 40 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); ra=d['roofline_all']
print('[$cfg] img/s %7.1f | ' % d['value'] + '  '.join('%s %.1f' % (k.replace('k_',''), v['avg_us']) for k,v in ra.items()))"
done
