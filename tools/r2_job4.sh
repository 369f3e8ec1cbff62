#!/bin/bash
# diagnostic counters for the fused kernel
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/diag; rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $OUT/avail.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -- python bench.py --no_cpu_baseline --no_latency --steps 1 --warmup 0 --prewarm_s 0 > /dev/null 2> $OUT/p$i.log
  python tools/rocpd_pmc.py $(find $OUT/p$i -name "*.db" | head -1) 2>&1 | grep -E "supgram|evd_kernelILi1|evd_kernelILi2|counter" | cut -c1-200 > $OUT/p$i.txt
  rm -rf $OUT/p$i
  echo "== $C"; cat $OUT/p$i.txt
done
