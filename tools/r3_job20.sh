mkdir -p gpurun_out/r3_20
timeout 900 python tools/bench_supgram.py > gpurun_out/r3_20/supgram_ablate.jsonl 2> gpurun_out/r3_20/err.log; cat gpurun_out/r3_20/supgram_ablate.jsonl; tail -3 gpurun_out/r3_20/err.log
