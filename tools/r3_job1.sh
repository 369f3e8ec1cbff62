# round 3, job 1: new concurrency / sharded tests, the stream-group probe, a bench line of the unchanged kernels
mkdir -p gpurun_out/r3_1
timeout 1500 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_sharded.py tests/test_gpu_lowrank_forward.py -x -q -m gpu > gpurun_out/r3_1/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_1/tests.log
tail -15 gpurun_out/r3_1/tests.log
timeout 600 python tools/repro_stream_groups.py --batch 16 --groups 2 --reps 6 > gpurun_out/r3_1/groups2.jsonl 2> gpurun_out/r3_1/groups2.err; tail -3 gpurun_out/r3_1/groups2.err; cat gpurun_out/r3_1/groups2.jsonl | cut -c1-400
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r3_1/bench.json 2> gpurun_out/r3_1/bench.err; tail -2 gpurun_out/r3_1/bench.err; cut -c1-600 gpurun_out/r3_1/bench.json
timeout 300 python tools/ref_gpu_baseline.py --reps 3 > gpurun_out/r3_1/ref_gpu.json 2> gpurun_out/r3_1/ref_gpu.err; cat gpurun_out/r3_1/ref_gpu.json; tail -2 gpurun_out/r3_1/ref_gpu.err
