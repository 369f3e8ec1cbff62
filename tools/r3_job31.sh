mkdir -p gpurun_out/r3_31
O=gpurun_out/r3_31
timeout 300 python tools/bench_evd_wave.py 2>/dev/null | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_evd_wave.py tests/test_gpu_twolevel.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for cfg in "4096 1" "768 16"; do
  set -- $cfg
  for q in 0 1; do timeout 300 python tools/check_evdq.py run $q $1 $2 $O/r_$1_$2_$q.npz 2>&1 | grep evdq | cut -c1-100; done
  python tools/check_evdq.py cmp $O/r_$1_$2_0.npz $O/r_$1_$2_1.npz
done
rm -f $O/*.npz
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()})"
