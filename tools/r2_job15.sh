#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 600 python -m pytest tests/test_gpu_svd.py -x -q -k "4096 or reduce or chol or tall" 2>&1 | tail -3 | cut -c1-300
for sb in 16 32; do
  timeout 600 python tools/full_model_bench.py --model llama-2-7b --no_parity --svd_batch $sb 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('7B svd_batch $sb', round(r['decompose_total_s'],2), r['sweeps_min_max'], round(r['max_mem_GB'],1))"
done
timeout 600 python tools/full_model_bench.py --model llama-2-13b --no_parity --svd_batch 32 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('13B svd_batch 32', round(r['decompose_total_s'],2), r['sweeps_min_max'], round(r['max_mem_GB'],1))"
for b in 16 32; do
python bench.py --no_cpu_baseline --no_latency --steps 3 --warmup 1 --prewarm_s 3 --batch $b > gpurun_out/bb_$b.json 2>/dev/null
python - gpurun_out/bb_$b.json $b <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("batch", sys.argv[2], "| SVD/s", round(r["value"], 2), "ms/step", round(r["ms_per_step"], 1), {k: (round(v["ms_per_step"], 1), v["launches"]) for k, v in r["roofline"]["classes"].items()})
PY
done
