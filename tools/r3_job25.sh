mkdir -p gpurun_out/r3_25
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -m gpu -k "reduction or llm_like or headline or mlp_shapes_sigma or 13b or rank_deficient" > gpurun_out/r3_25/t.log 2>&1; tail -3 gpurun_out/r3_25/t.log
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3_25
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python bench.py --no_cpu_baseline --no_latency --steps 2 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt
rm -rf $OUT/kt
grep -i "gram64\|chol\|permute\|r_to_f32" $OUT/kernel_stats.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_25/bench.json 2> gpurun_out/r3_25/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3_25/bench.json')); print(d['value'], d['ms_per_step'], d['config'].get('latency_batch1_ms'), {k:(round(v['ms_per_step'],1), v['launches']) for k,v in d['roofline']['classes'].items()}, d['roofline']['sweeps'][:4])"
