#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
for cfg in "ASVD_SUPER_GROUPED=0" "ASVD_SUPER_GROUPED=1"; do
  env $cfg timeout 600 python tools/full_model_bench.py --model llama-2-13b 2>gpurun_out/f13.err | tail -1 > gpurun_out/f13_$cfg.json
  python - "$cfg" gpurun_out/f13_$cfg.json <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "13B", round(r["decompose_total_s"], 2), r["sweeps_min_max"], [(p["shape"], p["sweeps"], "%.1e" % p["sigma_rel_err_top_r"], "%.1e" % p["orthogonality_max"]) for p in r["parity"]])
except Exception as e: print(sys.argv[1], "failed", e); print(open("gpurun_out/f13.err").read()[-600:])
PY
done
ASVD_SUPER_GROUPED=1 timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -k "13b" 2>&1 | grep -E "passed|failed" | tail -2
