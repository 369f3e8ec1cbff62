mkdir -p gpurun_out/r3_30
O=gpurun_out/r3_30
timeout 1500 python -m pytest tests/test_gpu_svd.py tests/test_gpu_concurrency.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
for la in 0 1; do
ASVD_CHOL_LOOKAHEAD=$la timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/bench_$la.json 2> $O/bench_$la.err; python -c "
import json; d=json.load(open('$O/bench_$la.json')); print($la, d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()})"
done
