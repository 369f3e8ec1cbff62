#!/bin/bash
# quick GPU check of a K4 change: SVD parity tests, then the bench with the two-level sweep on and off
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q 2>&1 | tail -15 > gpurun_out/check_tests.txt
cat gpurun_out/check_tests.txt
ASVD_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --prewarm_s 0 --no_cpu_baseline > gpurun_out/bench_dbg.json 2> gpurun_out/bench_dbg.err
grep "sweep .* wall" gpurun_out/bench_dbg.err | tail -12
timeout 600 python bench.py --steps 3 --warmup 1 --prewarm_s 3 > gpurun_out/bench_two.json 2> gpurun_out/bench_two.err
ASVD_TWOLEVEL=0 timeout 600 python bench.py --steps 3 --warmup 1 --prewarm_s 3 --no_cpu_baseline > gpurun_out/bench_one.json 2> gpurun_out/bench_one.err
python - <<'PY'
import json
for f in ("bench_two", "bench_one"):
    try:
        r = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, "SVD/s", round(r["value"], 2), "sweeps", r["roofline"]["sweeps"][:4], "sweep_ms", [round(x, 1) for x in r["roofline"]["sweep_wall_ms"]],
              "parity", r.get("parity"))
        print("   classes", {k: (round(v["ms_per_step"], 1), v["launches"]) for k, v in r["roofline"]["classes"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
