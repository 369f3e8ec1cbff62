#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 900 python tools/full_model_bench.py --model llama-2-7b 2>gpurun_out/full7b.err | tail -1 > gpurun_out/r2_full_7b.json
timeout 1200 python tools/full_model_bench.py --model llama-2-13b 2>gpurun_out/full13b.err | tail -1 > gpurun_out/r2_full_13b.json
timeout 900 python tools/full_model_bench.py --model llama-2-7b --full_rank --no_parity 2>gpurun_out/full7b_fr.err | tail -1 > gpurun_out/r2_full_7b_fullrank.json
for f in gpurun_out/r2_full_7b.json gpurun_out/r2_full_13b.json gpurun_out/r2_full_7b_fullrank.json; do python - $f <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print(sys.argv[1], {k: r[k] for k in ("linears", "factorize_s", "truncate_split_s", "decompose_total_s", "achieved_TFLOPs_full_svd_count", "sweeps_min_max", "max_mem_GB")})
    for p in r["parity"]: print("   ", p)
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
timeout 1500 python tools/cpu_baseline_full.py --out gpurun_out/r2_cpu_full_model.json 2>gpurun_out/cpu_full.err | tail -c 1500
timeout 900 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_r2c.json") if l.startswith("{")][-1])
print("bench", r["value"], r["ms_per_step"], r["step_wall_ms"], r["roofline"]["frac"], r["roofline"]["svd_level"]["frac"], r["cpu_baseline"]["value"], r.get("latency_batch1_ms"))
PY
bash tools/prof_final.sh r2c > gpurun_out/prof_r2c.log 2>&1; tail -12 gpurun_out/prof_r2c.log | cut -c1-200
