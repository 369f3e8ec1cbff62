#!/bin/bash
mkdir -p gpurun_out
export ASVD_STRICT=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; tail -c 1500 gpurun_out/bench_r2b.json
bash tools/prof_final.sh r2b > gpurun_out/prof_r2b.log 2>&1; tail -30 gpurun_out/prof_r2b.log | cut -c1-220
