#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 1200 python -m pytest tests/test_gpu_svd.py -x -q 2>&1 | tail -3 | cut -c1-300
bash tools/r2_exp.sh "ASVD_X=1" "ASVD_X=2"
