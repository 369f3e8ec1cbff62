mkdir -p gpurun_out/r3_17
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu -k "reduction or llm_like or headline or mlp_shapes_sigma" > gpurun_out/r3_17/t.log 2>&1; tail -3 gpurun_out/r3_17/t.log
for cg in 1 2 4 8; do
ASVD_CHOL_GROUP=$cg timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_17/bench_$cg.json 2> gpurun_out/r3_17/bench_$cg.err; python -c "
import json; d=json.load(open('gpurun_out/r3_17/bench_$cg.json')); print($cg, d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
done
