#!/bin/bash
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -k "4096 or batched or determin or reduce or tall or wide" 2>&1 | grep -E "passed|failed|Error" | tail -3 | cut -c1-300
for b in 1 32; do
python bench.py --no_cpu_baseline --no_latency --batch $b --steps 4 --warmup 2 --prewarm_s 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('batch $b', round(r['value'],2), round(r['ms_per_step'],1), {k:round(v['ms_per_step'],1) for k,v in r['roofline']['classes'].items() if k in ('pack','evd','supgram')})"
done
