#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4
bash tools/r2_exp.sh "ASVD_X=1" "ASVD_GRAM_SPLIT=1"
for extra in "" "--no_fused_ratios"; do
  timeout 900 python tools/gpu_e2e_cli.py opt-125m 16 $extra 2>/dev/null | tail -1 > gpurun_out/e2e_opt_${extra:2}.json
  python - gpurun_out/e2e_opt_${extra:2}.json "$extra" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read())
print("opt-125m", sys.argv[2] or "fused_ratios", {k: round(v, 1) for k, v in r["timings_s"].items()}, "ppl_after", r["ppl_after"], r["trace_tail"][-2:])
PY
done
timeout 1500 python tools/gpu_e2e_cli.py llama-2-7b 32 2>/dev/null | tail -1 > gpurun_out/e2e_7b_fused_ratios.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/e2e_7b_fused_ratios.json").read())
print("llama-2-7b n_calib 32 fused ratios", {k: round(v, 1) for k, v in r["timings_s"].items()}, "ppl_after", r["ppl_after"], r["trace_tail"][-2:])
PY
