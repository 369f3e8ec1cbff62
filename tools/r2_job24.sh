#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
ASVD_DEBUG_WORKFILL=255 timeout 1500 python -m pytest tests/test_gpu_svd.py tests/test_gpu_kernels.py tests/test_gpu_twolevel.py -x -q -k "not 13b and not lm_head" 2>&1 | tail -3 | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench_r2f.json 2> gpurun_out/bench_r2f.err; python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r2f.json") if l.startswith("{")][-1])
print("bench", r["value"], r["ms_per_step"], "roofline", r["roofline"]["frac"], r["roofline"]["traffic"], "svd_level", r["roofline"]["svd_level"]["frac"], "cpu", r["cpu_baseline"]["value"], "lat1", r.get("latency_batch1_ms"))
PY
