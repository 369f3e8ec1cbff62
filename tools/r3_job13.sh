mkdir -p gpurun_out/r3_13
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu -k "llama13b" > gpurun_out/r3_13/t13b.log 2>&1; tail -3 gpurun_out/r3_13/t13b.log
for G in 1 0; do
ASVD_SUPGRAM_GROUPED=$G timeout 600 python bench.py --m 5120 --n 5120 --batch 16 --steps 3 --warmup 1 --no_cpu_baseline --no_latency > gpurun_out/r3_13/bench5120_g$G.json 2> gpurun_out/r3_13/b.err; python -c "
import json; d=json.load(open('gpurun_out/r3_13/bench5120_g$G.json')); print('grouped-fused', $G, d['value'], d['ms_per_step'], {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4], d['roofline']['svd_level']['frac'])"
done
ASVD_SUPGRAM_GROUPED=1 timeout 900 python tools/full_model_bench.py --model llama-2-13b > gpurun_out/r3_13/full13b.json 2> gpurun_out/r3_13/full13b.err; tail -c 600 gpurun_out/r3_13/full13b.json; tail -2 gpurun_out/r3_13/full13b.err
