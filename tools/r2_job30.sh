#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_r2g.json 2> gpurun_out/bench_r2g.err; python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r2g.json") if l.startswith("{")][-1])
print("bench", r["value"], r["ms_per_step"], "roofline", r["roofline"]["frac"], r["roofline"]["traffic"], "svd_level", r["roofline"]["svd_level"]["frac"], "cpu", r["cpu_baseline"]["value"], "lat1", r.get("latency_batch1_ms"))
PY
