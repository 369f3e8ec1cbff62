mkdir -p gpurun_out/r3_27
O=gpurun_out/r3_27
for cfg in "4096 1" "4096 4" "768 16" "5120 2"; do
  set -- $cfg
  for q in 0 1; do timeout 300 python tools/check_evdq.py run $q $1 $2 $O/r_$1_$2_$q.npz 2>&1 | grep evdq; done
  python tools/check_evdq.py cmp $O/r_$1_$2_0.npz $O/r_$1_$2_1.npz
done
rm -f $O/*.npz
timeout 300 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_evd_wave.py -x -q -m gpu 2>&1 | tail -2
bash tools/r2_job14.sh 2>&1 | tee $O/shapes.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'))"
