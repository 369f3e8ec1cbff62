mkdir -p gpurun_out/r3_16
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu > gpurun_out/r3_16/t.log 2>&1; tail -3 gpurun_out/r3_16/t.log
for tile in 128 256; do
ASVD_SNAPSHOT_TILE=$tile timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_16/bench_$tile.json 2> gpurun_out/r3_16/bench_$tile.err; python -c "
import json; d=json.load(open('gpurun_out/r3_16/bench_$tile.json')); print($tile, d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
done
