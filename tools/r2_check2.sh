#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
ASVD_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_svd.py -x -q 2>&1 | tail -8
for cfg in "ASVD_SPLIT=1" "ASVD_SPLIT=1 ASVD_DUP2=1" "ASVD_SPLIT=0 ASVD_DUP2=1"; do
  env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --prewarm_s 3 > gpurun_out/c2.json 2> gpurun_out/c2.err
  python - "$cfg" gpurun_out/c2.json <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], "| SVD/s", round(r["value"], 2), "sweeps", r["roofline"]["sweeps"][:3], "sweep_ms", [round(x, 1) for x in r["roofline"]["sweep_wall_ms"]])
    print("     classes", {k: (round(v["ms_per_step"], 1), v["launches"]) for k, v in r["roofline"]["classes"].items()})
    print("     parity", r.get("parity"))
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("gpurun_out/c2.err").read()[-1500:])
PY
done
