#!/bin/bash
# A/B runs of the bench under different env knobs: bash tools/r2_exp.sh "ASVD_GROUPS=4" "ASVD_DUP2=3" ...
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --prewarm_s 3 --no_cpu_baseline > gpurun_out/exp_$i.json 2> gpurun_out/exp_$i.err
  python - "$cfg" gpurun_out/exp_$i.json <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], "| SVD/s", round(r["value"], 2), "sweeps", r["roofline"]["sweeps"][:3], "sweep_ms", [round(x, 1) for x in r["roofline"]["sweep_wall_ms"]])
    print("     parity", r.get("parity"), "lat1", r.get("latency_batch1_ms"))
    print("     classes", {k: (round(v["ms_per_step"], 1), v["launches"]) for k, v in r["roofline"]["classes"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
