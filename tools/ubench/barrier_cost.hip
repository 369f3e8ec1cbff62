// What does a workgroup barrier cost?  Loop of N __syncthreads() with nothing in between (and, second mode, with one LDS write + read around
// each), for the two shapes the library's streaming kernels use: 256 threads x 3 workgroups per CU (48 KB LDS each, snapshot kernel) and
// 512 threads x 1 workgroup per CU (156 KB, fused update + Gram kernel).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/barrier_cost.hip -o tools/bin/barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float* out, int n) {
    extern __shared__ float sm[];
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        if constexpr (MODE == 1) sm[threadIdx.x] = acc + (float)i;
        __syncthreads();
        if constexpr (MODE == 1) acc += sm[(threadIdx.x + 64) % blockDim.x];
        __syncthreads();
        asm volatile("" ::: "memory");
    }
    if (acc == 12345.0f) out[blockIdx.x] = acc;
}

template <int MODE>
static double run(int threads, int lds_bytes, int blocks, int n, float* out) {
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<blocks, threads, lds_bytes>>>(out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<MODE><<<blocks, threads, lds_bytes>>>(out, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* out; (void)hipMalloc(&out, 1 << 20);
    const int n = 4096;
    struct { int threads, lds, blocks; const char* name; } cfg[] = {
        {256, 48 * 1024, 768, "256 threads, 3 workgroups per CU"}, {256, 48 * 1024, 256, "256 threads, 1 workgroup per CU"},
        {512, 156 * 1024, 256, "512 threads, 1 workgroup per CU"}, {128, 33 * 1024, 1024, "128 threads, 4 workgroups per CU"},
        {512, 108 * 1024, 256, "512 threads (latency-form solver), 1 workgroup per CU"}};
    for (auto& c : cfg) {
        const double m0 = run<0>(c.threads, c.lds, c.blocks, n, out), m1 = run<1>(c.threads, c.lds, c.blocks, n, out);
        printf("{\"shape\": \"%s\", \"ns_per_barrier_empty\": %.1f, \"ns_per_barrier_with_lds_write_read\": %.1f}\n", c.name, 1e6 * m0 / (2.0 * n), 1e6 * m1 / (2.0 * n));
    }
    return 0;
}
