#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 600 python -m pytest tests/test_gpu_lowrank_forward.py -x -q 2>&1 | tail -3 | cut -c1-400
timeout 600 python tools/bench_aux.py > gpurun_out/r2_aux.jsonl 2> gpurun_out/aux.err; grep -E "lowrank" gpurun_out/r2_aux.jsonl | cut -c1-200
cd /tmp; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/auxprof
rocprofv3 --kernel-trace --stats -d gpurun_out/auxprof -- python tools/bench_aux.py > /dev/null 2> gpurun_out/aux2.err
python tools/rocpd_stats.py $(find gpurun_out/auxprof -name "*.db" | head -1) > gpurun_out/r2_aux_kernel_stats.txt
grep -E "^kernel|lowrank|Cijk|gemv|absstat" gpurun_out/r2_aux_kernel_stats.txt | cut -c1-170 | head -30
rm -rf gpurun_out/auxprof
