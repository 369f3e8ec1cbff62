# usage: bash tools/batch_sweep.sh "<groups>:<batch> ..."   e.g. "2:16 3:18"
for gb in ${1:-"0:1 0:2 0:4 0:8 0:16"}; do
  g=${gb%%:*}; b=${gb##*:}
  if [ "$g" != "0" ]; then export ASVD_GROUPS=$g; else unset ASVD_GROUPS; fi
  python bench.py --batch $b --steps 2 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); c=d['roofline']['classes']
print('groups', '$g', 'batch', $b, 'SVD/s %.2f'%d['value'], {k:(round(v['avg_us'],1), v['launches']) for k,v in c.items()}, d['roofline']['sweeps'][:3])
"
done
