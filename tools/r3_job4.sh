mkdir -p gpurun_out/r3_4
timeout 300 python tools/bench_evd_wave.py > gpurun_out/r3_4/evdw_micro.jsonl 2> gpurun_out/r3_4/evdw_micro.err; cat gpurun_out/r3_4/evdw_micro.jsonl; tail -2 gpurun_out/r3_4/evdw_micro.err
timeout 600 python tools/repro_two_streams.py --rounds 3 --only "zzz" --extra "evdw=0:ASVD_EVDW=0" --extra "evdw=0,ldszero:ASVD_EVDW=0,ASVD_FENCE=4" > gpurun_out/r3_4/two_streams.jsonl 2> gpurun_out/r3_4/two_streams.err
python - <<'PY'
import json
for l in open('gpurun_out/r3_4/two_streams.jsonl'):
    d=json.loads(l)
    print(d['config'], 'differ', d['n_differ'], '/', d['n_runs'], 'max_rel %.2e' % max(r['max_rel_dS'] for r in d['runs']), 'sweeps', [r['sweeps'] for r in d['runs']][:2])
PY
timeout 600 python tools/repro_stream_groups.py --batch 16 --groups 2 --reps 4 > gpurun_out/r3_4/groups2.jsonl 2> gpurun_out/r3_4/groups2.err; cut -c1-330 gpurun_out/r3_4/groups2.jsonl
