mkdir -p gpurun_out/r3_22
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3_22
cd $R
for a in 0 63 7 54; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/p_$a -- python tools/bench_supgram.py $a > $OUT/run_$a.log 2>&1
  python tools/rocpd_pmc.py $(find $OUT/p_$a -name "*.db" | head -1) 2>&1 | grep -i "supgram\|kernel " > $OUT/clk_$a.txt
  rm -rf $OUT/p_$a
  echo "== ablate $a"; grep ablate $OUT/run_$a.log; cat $OUT/clk_$a.txt
done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/p_bench -- python bench.py --no_cpu_baseline --no_latency --steps 1 --warmup 0 --prewarm_s 0 > /dev/null 2> $OUT/bench.log
python tools/rocpd_pmc.py $(find $OUT/p_bench -name "*.db" | head -1) > $OUT/clk_bench.txt 2>&1; rm -rf $OUT/p_bench; cat $OUT/clk_bench.txt
