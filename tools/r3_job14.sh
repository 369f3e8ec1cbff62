mkdir -p gpurun_out/r3_14
for B in 16 32; do
ASVD_SUPGRAM_GROUPED=1 timeout 600 python bench.py --m 5120 --n 5120 --batch $B --steps 3 --warmup 1 --no_cpu_baseline --no_latency > gpurun_out/r3_14/bench5120_b$B.json 2> gpurun_out/r3_14/b.err; python -c "
import json; d=json.load(open('gpurun_out/r3_14/bench5120_b$B.json')); print('batch', $B, d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4], d['roofline']['svd_level']['frac'])"
done
