#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 1500 python tools/gpu_e2e_cli.py llama-2-7b 32 2>gpurun_out/e2e7b.err | tail -1 > gpurun_out/e2e_7b_r2e.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/e2e_7b_r2e.json").read())
print("llama-2-7b n_calib 32", {k: round(v, 1) for k, v in r["timings_s"].items()}, "total", round(sum(r["timings_s"].values()), 1), "ppl_after", r["ppl_after"], r["trace_tail"][-2:])
PY
