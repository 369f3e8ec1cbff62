mkdir -p gpurun_out/r3_6
for L in 65536 51200; do
ASVD_EVDW_LDS=$L timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_latency > gpurun_out/r3_6/bench_lds$L.json 2> gpurun_out/r3_6/bench_lds$L.err; tail -1 gpurun_out/r3_6/bench_lds$L.err; python -c "
import json; d=json.load(open('gpurun_out/r3_6/bench_lds$L.json')); print($L, d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
done
timeout 900 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_sharded.py tests/test_gpu_export.py tests/test_gpu_lowrank_forward.py -x -q -m gpu > gpurun_out/r3_6/tests.log 2>&1; tail -8 gpurun_out/r3_6/tests.log
