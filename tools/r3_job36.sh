mkdir -p gpurun_out/r3_36
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r3_36
cd $R
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p_$i -- python bench.py --no_cpu_baseline --no_latency --steps 1 --warmup 0 --prewarm_s 0 > /dev/null 2> $OUT/p_$i.log
  python tools/rocpd_pmc.py $(find $OUT/p_$i -name "*.db" | head -1) 200 > $OUT/pmc_$i.txt 2>&1
  rm -rf $OUT/p_$i
  echo "== $C"; grep -i "fullcheck\|supgram\|evdw12\|kernel " $OUT/pmc_$i.txt | cut -c1-140
done
