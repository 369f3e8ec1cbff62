mkdir -p gpurun_out/r3_26
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_26/full.log 2>&1; tail -3 gpurun_out/r3_26/full.log
bash tools/r2_job14.sh 2>&1 | tee gpurun_out/r3_26/shapes.txt
