#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_lowrank_forward.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 600 python tools/bench_aux.py > gpurun_out/r2_aux.jsonl 2> gpurun_out/aux.err; grep -E "lowrank|absstat_abs_mean" gpurun_out/r2_aux.jsonl | cut -c1-160
