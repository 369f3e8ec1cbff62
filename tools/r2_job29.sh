#!/bin/bash
export ASVD_STRICT=1
for b in 1 2 4 8 16; do
python bench.py --no_cpu_baseline --no_latency --batch $b --steps 4 --warmup 2 --prewarm_s 2 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('batch $b', round(r['value'],2), round(r['ms_per_step'],1), {k:round(v['ms_per_step'],1) for k,v in r['roofline']['classes'].items() if k in ('evd','supgram')})"
done
timeout 600 python tools/full_model_bench.py --model llama-2-7b --no_parity 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('7B', round(r['decompose_total_s'],2), r['sweeps_min_max'])"
timeout 900 python -m pytest tests/test_gpu_svd.py tests/test_gpu_pipeline.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
