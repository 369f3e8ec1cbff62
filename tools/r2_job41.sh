#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 600 python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 > gpurun_out/e2e_opt_final.json
python -c "
import json; r=json.loads(open('gpurun_out/e2e_opt_final.json').read()); print('opt-125m', {k: round(v,1) for k,v in r['timings_s'].items()}, r['ppl_after'], r['trace_tail'][-2:])"
timeout 600 python tools/full_model_bench.py --model llama-2-7b --stable_rank 2>/dev/null | tail -1
timeout 600 python tools/full_model_bench.py --model opt-125m --no_parity 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('opt-125m decomposition', round(r['decompose_total_s'],3), r['sweeps_min_max'])"
