mkdir -p gpurun_out/r3_8
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu -k "llm_like or half_inputs or rank_deficient or batched" > gpurun_out/r3_8/svd.log 2>&1; tail -5 gpurun_out/r3_8/svd.log
ASVD_EVDW_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_8/bench.json 2> gpurun_out/r3_8/bench.err; grep "evdw12 trace" gpurun_out/r3_8/bench.err | head -4; python -c "
import json; d=json.load(open('gpurun_out/r3_8/bench.json')); print(d['value'], d['ms_per_step'], d.get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
