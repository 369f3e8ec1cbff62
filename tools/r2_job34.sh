#!/bin/bash
mkdir -p gpurun_out
run() {
  python bench.py --no_cpu_baseline --no_latency --steps 3 --warmup 1 --prewarm_s 3 "${@:2}" 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', '| SVD/s', round(r['value'],2), 'ms/step', round(r['ms_per_step'],1), 'sweeps', sorted(set(r['roofline']['sweeps'])), 'svd_level', round(r['roofline']['svd_level']['frac'],3))"
}
run 11008x4096_b8 --m 11008 --n 4096 --batch 8
run 11008x4096_b32 --m 11008 --n 4096 --batch 32
run 4096x11008_b8 --m 4096 --n 11008 --batch 8
run 5120_b16 --m 5120 --n 5120 --batch 16
run 5120_b32 --m 5120 --n 5120 --batch 32
run 13824x5120_b16 --m 13824 --n 5120 --batch 16
