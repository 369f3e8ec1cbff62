mkdir -p gpurun_out/r3_24
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3_24
cd $R
for a in 0 1 2 4 3 5 6 7; do
  ASVD_FC_ABLATE=$a rocprofv3 --kernel-trace --stats -d $OUT/kt_$a -- python bench.py --no_cpu_baseline --no_latency --steps 1 --warmup 0 --prewarm_s 0 > $OUT/b_$a.json 2> $OUT/kt_$a.log
  python tools/rocpd_stats.py $(find $OUT/kt_$a -name "*.db" | head -1) > $OUT/ks_$a.txt; rm -rf $OUT/kt_$a
  echo "ablate $a: $(grep fullcheck $OUT/ks_$a.txt)"
done
