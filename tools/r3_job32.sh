mkdir -p gpurun_out/r3_32
for i in 1 2; do python tools/full_model_bench.py --model llama-2-13b 2>/dev/null | tail -1 > gpurun_out/r3_32/full_13b_$i.json; python -c "
import json; x=json.load(open('gpurun_out/r3_32/full_13b_$i.json')); print(x['factorize_s'], x['decompose_total_s'], x['sweeps_min_max'])"; done
python bench.py --m 5120 --n 5120 --batch 32 --no_cpu_baseline --no_latency --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('5120 b32', d['value'], d['ms_per_step'])"
python bench.py --m 13824 --n 5120 --batch 16 --no_cpu_baseline --no_latency --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('13824x5120 b16', d['value'], d['ms_per_step'])"
