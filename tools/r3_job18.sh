mkdir -p gpurun_out/r3_18
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3_18
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python bench.py --no_cpu_baseline --no_latency --steps 2 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt
rm -rf $OUT/kt
head -40 $OUT/kernel_stats.txt
