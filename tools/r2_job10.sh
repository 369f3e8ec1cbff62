#!/bin/bash
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/auxprof
rocprofv3 --kernel-trace --stats -d gpurun_out/auxprof -- python tools/bench_aux.py > gpurun_out/r2_aux.jsonl 2> gpurun_out/aux.err
grep -E "lowrank|absstat_abs_mean|truncate_split" gpurun_out/r2_aux.jsonl | cut -c1-260
python tools/rocpd_stats.py $(find gpurun_out/auxprof -name "*.db" | head -1) > gpurun_out/r2_aux_kernel_stats.txt
grep -E "kernel|absstat|lowrank|truncate|Cijk|fro_|scale_cols" gpurun_out/r2_aux_kernel_stats.txt | cut -c1-170 | head -30
rm -rf gpurun_out/auxprof
