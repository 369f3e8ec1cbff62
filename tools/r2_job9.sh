#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -5 | cut -c1-300
timeout 600 python tools/bench_aux.py > gpurun_out/r2_aux.jsonl 2> gpurun_out/aux.err; cat gpurun_out/r2_aux.jsonl | cut -c1-200
bash tools/r2_exp.sh "ASVD_EVD_PAIRS=24" "ASVD_EVD_PAIRS=16" "ASVD_X=1"
