for v in ${1:-"1 2 3"}; do
  ASVD_INNER=$v python bench.py --batch 16 --steps 2 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); c=d['roofline']['classes']
print('inner $v', 'SVD/s %.2f'%d['value'], {k:(round(v['avg_us'],1), v['launches']) for k,v in c.items()}, d['roofline']['sweeps'][:4])
"
done
