#!/bin/bash
# Round-3 evidence on one MI355X (about 50 minutes): PMC + kernel-trace passes first, so that the bench line of the same library binary
# carries roofline.traffic; outputs under gpurun_out/final/ (copy what should be kept to profiles/).
set -u
O=gpurun_out/final; mkdir -p $O
export ASVD_STRICT=1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PMC_BATCH=32 bash tools/prof_final.sh r3 > $O/prof.log 2>&1
cp gpurun_out/prof_r3/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json
python tools/full_model_bench.py --model llama-2-7b 2>/dev/null | tail -1 > $O/full_7b.json
python tools/full_model_bench.py --model llama-2-13b 2>/dev/null | tail -1 > $O/full_13b.json
python tools/full_model_bench.py --model opt-125m 2>/dev/null | tail -1 > $O/full_opt125m.json
python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 > $O/e2e_opt125m.json
python tools/bench_aux.py > $O/aux.jsonl 2> /dev/null
python tools/ref_gpu_baseline.py > $O/ref_gpu.json 2> /dev/null
python tools/bench_evd_wave.py > $O/evdw_micro.jsonl 2> /dev/null
python tools/gpu_e2e_cli.py llama-2-7b 32 2>/dev/null | tail -1 > $O/e2e_llama2_7b_ncalib32.json
for f in full_7b full_13b full_opt125m e2e_opt125m e2e_llama2_7b_ncalib32; do echo "== $f"; head -c 400 $O/$f.json; echo; done
