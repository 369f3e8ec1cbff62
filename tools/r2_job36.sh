#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 600 python -m pytest tests/test_gpu_lowrank_forward.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 600 python tools/bench_aux.py > gpurun_out/r2_aux.jsonl 2> gpurun_out/aux.err; grep -E "lowrank" gpurun_out/r2_aux.jsonl | cut -c1-200
