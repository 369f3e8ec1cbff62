cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_q; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python bench.py --no_cpu_baseline --steps 1 --warmup 0 --prewarm_s 0 > /dev/null 2> $OUT/kt.log
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) | head -14
rm -rf $OUT/kt
