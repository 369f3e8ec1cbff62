#!/bin/bash
export ASVD_STRICT=1
ASVD_DEBUG_WORKFILL=255 timeout 1500 python -m pytest tests/test_gpu_svd.py tests/test_gpu_kernels.py tests/test_gpu_twolevel.py -x -q -k "not 13b and not lm_head and not rccl" 2>&1 | grep -E "passed|failed|Error" | tail -5 | cut -c1-300
