#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "absstat or hook" 2>&1 | tail -3 | cut -c1-300
for cfg in "ASVD_SUPER_RR=0" "ASVD_SUPER_RR=1"; do
  env $cfg timeout 600 python tools/full_model_bench.py --model llama-2-13b 2>gpurun_out/f13.err | tail -1 > gpurun_out/f13_$cfg.json
  python - "$cfg" gpurun_out/f13_$cfg.json <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); print(sys.argv[1], "13B", round(r["decompose_total_s"], 2), r["sweeps_min_max"], [(p["shape"], p["sweeps"], "%.1e" % p["sigma_rel_err_top_r"], "%.1e" % p["orthogonality_max"]) for p in r["parity"]])
except Exception as e: print(sys.argv[1], "failed", e); print(open("gpurun_out/f13.err").read()[-600:])
PY
done
ASVD_SUPER_RR=1 timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -k "13b" 2>&1 | tail -3 | cut -c1-300
ASVD_SUPER_RR=1 timeout 600 python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('opt rr', {k: round(v,1) for k,v in r['timings_s'].items()}, r['ppl_after'])"
timeout 600 python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('opt xor', {k: round(v,1) for k,v in r['timings_s'].items()}, r['ppl_after'])"
