#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r2d.json") if l.startswith("{")][-1])
print("bench", r["value"], r["ms_per_step"], r["step_wall_ms"], "roofline", r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["traffic"], "svd_level", r["roofline"]["svd_level"]["frac"], "cpu", r["cpu_baseline"]["value"], "lat1", r.get("latency_batch1_ms"), r.get("parity"))
PY
PMC_BATCH=32 bash tools/prof_final.sh r2d > gpurun_out/prof_r2d.log 2>&1; grep -A6 '"supgram"' gpurun_out/prof_r2d/pmc_traffic.json | head -8
timeout 900 python tools/full_model_bench.py --model llama-2-7b 2>gpurun_out/full7b.err | tail -1 > gpurun_out/r2_full_7b.json
timeout 1200 python tools/full_model_bench.py --model llama-2-13b 2>gpurun_out/full13b.err | tail -1 > gpurun_out/r2_full_13b.json
for f in gpurun_out/r2_full_7b.json gpurun_out/r2_full_13b.json; do python - $f <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print(sys.argv[1], {k: r[k] for k in ("linears", "svd_batch", "decompose_total_s", "achieved_TFLOPs_full_svd_count", "sweeps_min_max", "max_mem_GB")}, [("%.1e" % p["sigma_rel_err_top_r"], "%.1e" % p["orthogonality_max"]) for p in r["parity"]])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
python bench.py --no_cpu_baseline --no_latency --batch 16 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch16', r['value'], r['ms_per_step'])"
