#!/bin/bash
export ASVD_STRICT=1
timeout 600 python -m pytest tests/test_gpu_twolevel.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_svd.py -x -q -k "13b or 4096" 2>&1 | grep -E "passed|failed" | tail -2
