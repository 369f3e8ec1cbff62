mkdir -p gpurun_out/r3_12
timeout 300 python -m pytest tests/test_gpu_evd_wave.py tests/test_gpu_twolevel.py -x -q -m gpu > gpurun_out/r3_12/t.log 2>&1; tail -3 gpurun_out/r3_12/t.log
timeout 300 python tools/bench_evd_wave.py > gpurun_out/r3_12/evdw_micro.jsonl 2> gpurun_out/r3_12/evdw_micro.err; cut -c1-200 gpurun_out/r3_12/evdw_micro.jsonl
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_12/bench.json 2> gpurun_out/r3_12/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3_12/bench.json')); print(d['value'], d['ms_per_step'], d.get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
