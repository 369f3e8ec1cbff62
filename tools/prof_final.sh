# rocprofv3 evidence for the round's final bench command: kernel trace + stats, then PMC passes in their OWN runs
# (--kernel-trace only, one counter group per pass: FETCH_SIZE, WRITE_SIZE, MFMA / busy cycles).  Usage: bash tools/prof_final.sh <tag>
TAG=${1:-r4}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python bench.py --no_cpu_baseline --no_latency $BENCH_ARGS > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt
python tools/rocpd_overlap.py $DB >> $OUT/kernel_stats.txt 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -- python bench.py --no_cpu_baseline --no_latency --steps 1 --warmup 0 --prewarm_s 0 $BENCH_ARGS > /dev/null 2> $OUT/pmc_$N.log
  python tools/rocpd_pmc.py $(find $OUT/pmc_$N -name "*.db" | head -1) 400 > $OUT/pmc_$N.txt 2>&1
  rm -rf $OUT/pmc_$N
done
python tools/pmc_to_json.py $OUT > $OUT/pmc_traffic.json 2> $OUT/pmc_to_json.err
rm -rf $OUT/kt
head -14 $OUT/kernel_stats.txt; for f in $OUT/pmc_*.txt; do echo "== $f"; head -9 $f; done; cat $OUT/pmc_traffic.json | head -40
