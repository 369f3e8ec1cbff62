#!/bin/bash
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -k "4096 or batched or determin or small" 2>&1 | grep -E "passed|failed" | tail -2
bash tools/r2_exp.sh "ASVD_X=1"
