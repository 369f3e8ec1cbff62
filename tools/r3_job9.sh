mkdir -p gpurun_out/r3_9
timeout 600 python tools/repro_two_streams.py --rounds 3 --only "evdw" --extra "evdw=0:ASVD_EVDW=0" --extra "evdw=0 ldszero:ASVD_EVDW=0,ASVD_FENCE=4" --extra "evdw=0 noprio:ASVD_EVDW=0,ASVD_FENCE=8" --extra "evdw=0 both:ASVD_EVDW=0,ASVD_FENCE=12" > gpurun_out/r3_9/two_streams.jsonl 2> gpurun_out/r3_9/two_streams.err
python - <<'PY'
import json
for l in open('gpurun_out/r3_9/two_streams.jsonl'):
    d=json.loads(l)
    print(d['config'], 'differ', d['n_differ'], '/', d['n_runs'], 'max_rel %.2e' % max(r['max_rel_dS'] for r in d['runs']), 'sweeps', [r['sweeps'] for r in d['runs']][:2])
PY
tail -2 gpurun_out/r3_9/two_streams.err
