#!/bin/bash
mkdir -p gpurun_out
run() { # tag, args
  python bench.py --no_cpu_baseline --no_latency --steps 3 --warmup 1 --prewarm_s 3 "${@:2}" > gpurun_out/shape_$1.json 2> gpurun_out/shape_$1.err
  python - gpurun_out/shape_$1.json "$1" <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "| SVD/s", round(r["value"], 2), "ms/step", round(r["ms_per_step"], 1), "sweeps", sorted(set(r["roofline"]["sweeps"])), "svd_level", round(r["roofline"]["svd_level"]["frac"], 3))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run b8 --batch 8
run b24 --batch 24
run b32 --batch 32
run 11008x4096_b8 --m 11008 --n 4096 --batch 8
run 4096x11008_b8 --m 4096 --n 11008 --batch 8
run 5120_b16 --m 5120 --n 5120 --batch 16
run 768_b16 --m 768 --n 768 --batch 16 --rank 345
