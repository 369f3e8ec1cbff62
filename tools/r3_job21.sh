mkdir -p gpurun_out/r3_21
timeout 600 python -m pytest tests/test_gpu_twolevel.py -x -q -m gpu > gpurun_out/r3_21/t.log 2>&1; tail -3 gpurun_out/r3_21/t.log
for a in 0 1; do timeout 300 python tools/bench_supgram.py $a; done 2>/dev/null
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_21/bench.json 2> gpurun_out/r3_21/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3_21/bench.json')); print(d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
