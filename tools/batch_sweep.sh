for b in 1 2 3 4 6 8; do
  python bench.py --batch $b --steps 2 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); c=d['roofline']['classes']
print('batch', $b, 'SVD/s %.2f'%d['value'], {k:(round(v['avg_us'],1), v['launches']) for k,v in c.items()}, d['roofline']['sweeps'][:3])
"
done
