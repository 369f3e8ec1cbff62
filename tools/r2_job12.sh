#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 600 python -m pytest tests/test_gpu_lowrank_forward.py tests/test_gpu_twolevel.py -x -q 2>&1 | tail -4 | cut -c1-300
for cfg in "ASVD_SUPGRAM_FILL=0" "ASVD_SUPGRAM_FILL=0.75" "ASVD_SUPGRAM_FILL=0.5" "ASVD_SUPGRAM=0" "ASVD_TWOLEVEL=0"; do
  env $cfg timeout 600 python tools/full_model_bench.py --model llama-2-13b --no_parity 2>gpurun_out/f13.err | tail -1 > gpurun_out/f13.json
  python - "$cfg" <<'PY'
import json, sys
try:
    r = json.load(open("gpurun_out/f13.json")); print(sys.argv[1], "13B", round(r["decompose_total_s"], 2), r["sweeps_min_max"])
except Exception as e: print(sys.argv[1], "failed", e)
PY
done
env ASVD_SUPGRAM_FILL=0.75 timeout 600 python tools/full_model_bench.py --model llama-2-7b --no_parity 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('7B', round(r['decompose_total_s'],2), r['sweeps_min_max'])"
timeout 600 python tools/bench_aux.py > gpurun_out/r2_aux.jsonl 2> gpurun_out/aux.err; grep -E "lowrank|absstat_abs_mean" gpurun_out/r2_aux.jsonl | cut -c1-230
