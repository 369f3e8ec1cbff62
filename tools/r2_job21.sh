#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 | cut -c1-300
python bench.py --no_cpu_baseline --no_latency --batch 1 --steps 5 --warmup 2 --prewarm_s 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('batch1', r['value'], r['ms_per_step'], {k:(round(v['ms_per_step'],2),v['launches'],round(v['avg_us'],1)) for k,v in r['roofline']['classes'].items()}, r['roofline']['sweep_wall_ms'])"
python bench.py --no_cpu_baseline --no_latency --batch 4 --steps 5 --warmup 2 --prewarm_s 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('batch4', r['value'], r['ms_per_step'], {k:(round(v['ms_per_step'],2),v['launches'],round(v['avg_us'],1)) for k,v in r['roofline']['classes'].items()})"
