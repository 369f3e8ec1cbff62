#!/bin/bash
# Reproduces the evidence on one MI355X (round 6, final library: tools/evidence_r6_final.sh is the shorter script the committed profiles/r6_* came from).  Outputs under gpurun_out/; copy what should be kept to profiles/ (named r<round>_*).
#   bash tools/reproduce_evidence.sh            # everything (about 60 minutes)
#   bash tools/reproduce_evidence.sh quick      # tests + smoke + bench only (about 20 minutes)
#   bash tools/reproduce_evidence.sh prof       # rocprofv3 kernel stats + PMC passes + power telemetry + bench (about 12 minutes): run this one
#                                               # FIRST after a library change: it writes profiles/pmc_traffic.json's source, which bench.py quotes
#                                               # as roofline.traffic only when it was collected from the SAME libasvd_hip.so (sha256)
set -u
mkdir -p gpurun_out
export ASVD_STRICT=1   # (the K10 wrapper reads its give-up word every 64th launch under STRICT since round 5: bench_aux.py is not distorted by it any more)
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
[ "${1:-all}" = prof ] || python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
[ "${1:-all}" = prof ] || { python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err && tail -c 600 gpurun_out/bench.json; }
[ "${1:-all}" = quick ] && exit 0
# rocprofv3 kernel stats + FETCH/WRITE/MFMA counter passes of the bench workload alone (no model leg), UNSPLIT: every kernel alone on the chip — what
# bench.py's `roofline` quotes (its profiled step is never split); then the kernel trace of the product default (two half-batch streams): overlap factor
# PMC_STEPS: steps of the workload inside one counter pass (`bench.py --steps 1 --warmup 0 --prewarm_s 0` under ASVD_SPLIT=0 = the timed step + the unsplit profiled
# step); PMC_TAG: the name the pmc_*.txt files get under profiles/ (pmc_traffic.json:source points there; tests/test_host_logic.py checks that it resolves)
BENCH_ARGS="--sharded_model none" ASVD_SPLIT=0 PMC_BATCH=32 PMC_STEPS=2 PMC_TAG=${PMC_TAG:-repro} bash tools/prof_final.sh repro > gpurun_out/prof_repro.log 2>&1
( cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/kt_split
  rocprofv3 --kernel-trace --stats -d gpurun_out/kt_split -- python bench.py --no_cpu_baseline --no_latency --sharded_model none --steps 3 --warmup 1 --prewarm_s 2 > /dev/null 2> gpurun_out/kt_split.log
  DB=$(find gpurun_out/kt_split -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB | head -24 > gpurun_out/kernel_stats_split.txt; python tools/rocpd_overlap.py $DB | tail -2 >> gpurun_out/kernel_stats_split.txt; rm -rf gpurun_out/kt_split )
for w in idle mfma supgram supgram_ni bench; do python tools/power_probe.py --workload $w --seconds 6 --out gpurun_out/power.jsonl > /dev/null 2>&1; done   # socket power / clock / cap
python tools/bench_supgram.py > gpurun_out/supgram_micro.jsonl 2> /dev/null
python tools/bench_families.py > gpurun_out/families.txt 2> /dev/null          # sweeps and SVD/s per input family (round 6)
python tools/sweep_gemm_ceiling.py > gpurun_out/sweep_gemm_ceiling.txt 2> /dev/null
[ "${1:-all}" = prof ] && exit 0
python tools/full_model_bench.py --model llama-2-7b 2>/dev/null | tail -1 > gpurun_out/full_7b.json
python tools/full_model_bench.py --model llama-2-13b 2>/dev/null | tail -1 > gpurun_out/full_13b.json
python tools/cpu_baseline_full.py --out gpurun_out/cpu_full_model.json > /dev/null 2>&1
python tools/bench_aux.py > gpurun_out/aux.jsonl 2> /dev/null
python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 > gpurun_out/e2e_opt125m.json
python tools/gpu_e2e_cli.py llama-2-7b 32 2>/dev/null | tail -1 > gpurun_out/e2e_llama2_7b_ncalib32.json   # ~13 minutes
ls -la gpurun_out | tail -20
