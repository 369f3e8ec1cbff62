#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 600 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_lowrank_forward.py -x -q 2>&1 | tail -15 | cut -c1-300
bash tools/r2_exp.sh "ASVD_SUPGRAM=0 ASVD_SPARSE_ROUNDS=0" "ASVD_SUPGRAM=0" "ASVD_X=1" "ASVD_SUPGRAM_CHUNKS=8" "ASVD_SUPGRAM_CHUNKS=2"
