mkdir -p gpurun_out/r3_11
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_11/full.log 2>&1; tail -12 gpurun_out/r3_11/full.log
