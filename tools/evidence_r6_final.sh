#!/bin/bash
# Round-6 evidence of the FINAL library (int8 Gram / long-side product / snapshot), one MI355X, ~30 minutes.  Outputs under gpurun_out/ev/; copied to
# profiles/ as r6_* (the e2e records keep the library they were measured with: they time PyTorch forwards, not these kernels).
set -u
export ASVD_STRICT=1
O=gpurun_out/ev; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 > $O/smoke.txt
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gpu_tests_tail.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_steps20.json 2> $O/bench_driver.err
ASVD_SPLIT=0 python bench.py --no_cpu_baseline --sharded_model none > $O/bench_nosplit.json 2>/dev/null
BENCH_ARGS="--sharded_model none" ASVD_SPLIT=0 PMC_BATCH=32 PMC_STEPS=2 PMC_TAG=r6 bash tools/prof_final.sh r6 > $O/prof_r6.log 2>&1
( cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/kt_split
  rocprofv3 --kernel-trace --stats -d gpurun_out/kt_split -- python bench.py --no_cpu_baseline --no_latency --sharded_model none --steps 3 --warmup 1 --prewarm_s 2 > $O/bench_under_rocprof_split.json 2> $O/kt_split.log
  DB=$(find gpurun_out/kt_split -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB | head -30 > $O/kernel_stats_split.txt; python tools/rocpd_overlap.py $DB | tail -2 >> $O/kernel_stats_split.txt; rm -rf gpurun_out/kt_split )
python tools/bench_families.py > $O/families.txt 2> /dev/null
for a in "--m 11008 --n 4096 --batch 32" "--m 4096 --n 11008 --batch 32" "--m 5120 --n 5120 --batch 32" "--m 13824 --n 5120 --batch 16" "--m 768 --n 768 --batch 16 --rank 345" "--batch 8" "--batch 16" "--m 2048 --n 2048 --batch 32"; do
  python bench.py $a --no_cpu_baseline --no_latency --sharded_model none --steps 3 --warmup 1 --prewarm_s 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$a', round(d['value'], 2), round(d['ms_per_step'], 1), round(r['svd_level'].get('frac_of_fp32_mfma_peak', r['svd_level'].get('frac', 0)), 3), r['sweeps'][:3])" >> $O/shapes.txt 2>> $O/shapes.err
done
python tools/full_model_bench.py --model llama-2-7b 2>/dev/null | tail -1 > $O/full_7b.json
python tools/full_model_bench.py --model llama-2-13b 2>/dev/null | tail -1 > $O/full_13b.json
python tools/bench_i8_gemm.py > $O/i8_gemm_micro.jsonl 2>/dev/null
python tools/bench_aux.py > $O/aux.jsonl 2> /dev/null
ls -la $O gpurun_out/prof_r6 | tail -40
# whole pipeline (asvd.py flags) on one GPU with the final library: BASELINE configs[2] (7B shapes, n_calib 32, ~13 min), configs[4]'s workload at n_calib 4, opt-125m-shaped
if [ "${1:-all}" = all ]; then
python tools/gpu_e2e_cli.py opt-125m 16 2>/dev/null | tail -1 > $O/e2e_opt125m.json
python tools/gpu_e2e_cli.py llama-2-13b 4 --param_ratio_target 0.95 2>/dev/null | tail -1 > $O/e2e_llama2_13b_ratio095_ncalib4.json
python tools/gpu_e2e_cli.py llama-2-7b 32 2>/dev/null | tail -1 > $O/e2e_llama2_7b_ncalib32.json
ls -la $O | tail -5
fi
# the remaining round-6 records (each was its own job):
#   for w in idle gram_i8 gram_fp64 mfma supgram bench; do python tools/power_probe.py --workload $w --seconds 6 --out gpurun_out/ev/power.jsonl; done          -> profiles/r6_power.jsonl
#   ASVD_DEBUG_WORKFILL=255 python -m pytest tests/test_gpu_svd.py tests/test_gpu_gram_i8.py tests/test_gpu_families.py tests/test_gpu_full_model.py -q   -> profiles/r6_poisoned_workspace_tests.txt
#   ASVD_GRAM_I8=0 ASVD_SNAP_I8=0 ASVD_NN_I8=0 python -m pytest tests/test_gpu_svd.py tests/test_gpu_families.py -q                                      -> profiles/r6_legacy_kernels_tests.txt
#   for v in 4 6 ...; do ASVD_CHOL_GROUP=$v python bench.py --no_cpu_baseline --no_latency --sharded_model none --steps 4 --warmup 1 --prewarm_s 3; done   -> profiles/r6_chol_group.txt
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 bench.py --gpus 2 --steps 3 --warmup 1 --dist_backend gloo --same_gpu -> profiles/r6_bench_2ranks_one_gpu_gloo.json
