# rocprofv3 evidence for the round's final bench command (kernel trace, then PMC passes in their own runs)
set -x
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_m; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python bench.py --no_cpu_baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt
python tools/rocpd_overlap.py $DB >> $OUT/kernel_stats.txt 2>/dev/null
find $OUT/kt -name "*stats*.csv" | head
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python bench.py --no_cpu_baseline --batch 16 --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- python bench.py --no_cpu_baseline --batch 16 --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_write.log
python tools/rocpd_pmc.py $(find $OUT/pmc_fetch -name "*.db" | head -1) > $OUT/pmc_fetch.txt 2>&1
python tools/rocpd_pmc.py $(find $OUT/pmc_write -name "*.db" | head -1) > $OUT/pmc_write.txt 2>&1
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write
head -12 $OUT/kernel_stats.txt; head -8 $OUT/pmc_fetch.txt; head -8 $OUT/pmc_write.txt
