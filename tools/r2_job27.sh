#!/bin/bash
for b in 1 4; do for c in 2 4 8 16; do
ASVD_SUPGRAM_CHUNKS=$c python bench.py --no_cpu_baseline --no_latency --batch $b --steps 5 --warmup 2 --prewarm_s 1 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('batch $b chunks $c', round(r['ms_per_step'],1), {k:round(v['ms_per_step'],1) for k,v in r['roofline']['classes'].items() if k in ('evd','supgram')})"
done; done
