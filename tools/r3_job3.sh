mkdir -p gpurun_out/r3_3
timeout 600 python -m pytest tests/test_gpu_evd_wave.py -x -q -m gpu > gpurun_out/r3_3/evdw.log 2>&1; tail -15 gpurun_out/r3_3/evdw.log
timeout 900 python tools/repro_two_streams.py --rounds 3 --only "threads" --extra "evdw=0:ASVD_EVDW=0" --extra "split=0:ASVD_SPLIT=0" --extra "twolevel=0:ASVD_TWOLEVEL=0" --extra "noreduce:ASVD_NO_REDUCE=1" --extra "sparse=0:ASVD_SPARSE=0" > gpurun_out/r3_3/two_streams.jsonl 2> gpurun_out/r3_3/two_streams.err
tail -3 gpurun_out/r3_3/two_streams.err
python - <<'PY'
import json
for l in open('gpurun_out/r3_3/two_streams.jsonl'):
    d=json.loads(l)
    print(d['config'], 'differ', d['n_differ'], '/', d['n_runs'], 'max_rel %.2e' % max(r['max_rel_dS'] for r in d['runs']), 'sweeps', [r['sweeps'] for r in d['runs']][:2], 'ref', d['ref_sweeps'][:1])
PY
timeout 1200 python -m pytest tests/test_gpu_svd.py tests/test_gpu_twolevel.py -x -q -m gpu > gpurun_out/r3_3/svd.log 2>&1; tail -15 gpurun_out/r3_3/svd.log
timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/r3_3/bench_evdw.json 2> gpurun_out/r3_3/bench_evdw.err; tail -2 gpurun_out/r3_3/bench_evdw.err; python -c "
import json; d=json.load(open('gpurun_out/r3_3/bench_evdw.json')); print(d['value'], d['ms_per_step'], d.get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
