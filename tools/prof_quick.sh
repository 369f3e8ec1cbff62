cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_q; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python bench.py --no_cpu_baseline --steps 1 --warmup 0 --prewarm_s 0 > /dev/null 2> $OUT/kt.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB | head -16
python tools/rocpd_overlap.py $DB | tail -2
python tools/rocpd_timeline.py $DB > $OUT/timeline.txt 2>&1; head -60 $OUT/timeline.txt
rm -rf $OUT/kt
