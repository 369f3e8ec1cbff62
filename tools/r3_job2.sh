mkdir -p gpurun_out/r3_2
timeout 900 python tools/repro_two_streams.py --rounds 3 > gpurun_out/r3_2/two_streams.jsonl 2> gpurun_out/r3_2/two_streams.err
tail -3 gpurun_out/r3_2/two_streams.err
python - <<'PY'
import json
for l in open('gpurun_out/r3_2/two_streams.jsonl'):
    d=json.loads(l)
    print(d['config'], 'differ', d['n_differ'], '/', d['n_runs'], 'max_rel', max(r['max_rel_dS'] for r in d['runs']), 'sweeps', [r['sweeps'] for r in d['runs']][:2], 'ref', d['ref_sweeps'][:1])
PY
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_lowrank_forward.py tests/test_gpu_concurrency.py::test_mixed_schedules_concurrently -x -q -m gpu > gpurun_out/r3_2/tests.log 2>&1; tail -12 gpurun_out/r3_2/tests.log
