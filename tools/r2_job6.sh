#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do
  python bench.py --no_cpu_baseline --no_latency --steps 4 --warmup 1 --prewarm_s 3 > gpurun_out/rep_$i.json 2> gpurun_out/rep_$i.err
  python - gpurun_out/rep_$i.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("rep", round(r["value"], 2), [round(x) for x in r["step_wall_ms"]], "prewarm", r["prewarm_steps"], "sweeps_ms", [round(x, 1) for x in r["roofline"]["sweep_wall_ms"]])
PY
done
ASVD_DEBUG=1 python bench.py --no_cpu_baseline --no_latency --steps 3 --warmup 1 --prewarm_s 3 2>&1 | grep -E "sweep [0-9]+.*wall|^\{" | cut -c1-120 | tail -45
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head
