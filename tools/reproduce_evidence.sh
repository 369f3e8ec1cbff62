#!/bin/bash
# Reproduces the round's evidence on one MI355X (about 45 minutes).  Round 3 used tools/r3_final.sh (same steps, PMC passes first so that the
# bench line carries roofline.traffic of the same library binary).  Outputs under gpurun_out/; copy what should be kept to profiles/.
#   bash tools/reproduce_evidence.sh            # everything
#   bash tools/reproduce_evidence.sh quick      # tests + smoke + bench only (about 10 minutes)
set -u
mkdir -p gpurun_out
export ASVD_STRICT=1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err && tail -c 600 gpurun_out/bench.json
[ "${1:-all}" = quick ] && exit 0
PMC_BATCH=32 bash tools/prof_final.sh repro > gpurun_out/prof_repro.log 2>&1   # rocprofv3 kernel stats + FETCH/WRITE/MFMA counter passes
python tools/full_model_bench.py --model llama-2-7b 2>/dev/null | tail -1 > gpurun_out/full_7b.json
python tools/full_model_bench.py --model llama-2-13b 2>/dev/null | tail -1 > gpurun_out/full_13b.json
python tools/cpu_baseline_full.py --out gpurun_out/cpu_full_model.json > /dev/null 2>&1
python tools/bench_aux.py > gpurun_out/aux.jsonl 2> /dev/null
python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 > gpurun_out/e2e_opt125m.json
python tools/gpu_e2e_cli.py llama-2-7b 32 2>/dev/null | tail -1 > gpurun_out/e2e_llama2_7b_ncalib32.json   # ~13 minutes
ls -la gpurun_out | tail -20
