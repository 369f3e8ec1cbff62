#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 900 python -m pytest tests/test_gpu_svd.py -x -q -k "4096 or mlp or batched" 2>&1 | tail -3 | cut -c1-300
bash tools/r2_exp.sh "ASVD_XCD_REMAP=0" "ASVD_XCD_REMAP=1"
