mkdir -p gpurun_out/r3_39
O=gpurun_out/r3_39
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu -k "headline or llm_like or nan or rank_deficient or values_only or mlp_shapes_full" 2>&1 | grep -E "passed|failed"
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()})"
