mkdir -p gpurun_out/r3_7
ASVD_EVDW_TRACE=1 ASVD_EVDW_LDS=51200 timeout 300 python bench.py --steps 1 --warmup 0 --prewarm_s 0 --no_cpu_baseline --no_latency > gpurun_out/r3_7/b.json 2> gpurun_out/r3_7/trace.err; grep "evdw12 trace" gpurun_out/r3_7/trace.err
