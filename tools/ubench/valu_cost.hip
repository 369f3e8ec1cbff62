// Issue cost of the instruction classes the wave-local eigen-solver is made of, one or two waves per SIMD, 64 independent registers per
// instruction class (no dependency stalls).  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/valu_cost.hip -o /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

template <int CTRL, bool BC>
__device__ __forceinline__ float dppf(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, BC)); }

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float c0) {
    float r[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) r[i] = (float)(threadIdx.x + i) * 1e-3f;
    float c = c0, s = 0.5f * c0;
    const int lane = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if constexpr (MODE == 0) r[i] = fmaf(r[i], c, s);                                       // v_fma / v_fmac
            if constexpr (MODE == 1) r[i] = c * dppf<0xB1, true>(r[i]);                             // v_mul_f32_dpp quad_perm
            if constexpr (MODE == 2) r[i] = c * dppf<0x130, true>(r[i]);                            // v_mul_f32_dpp wave_shl:1
            if constexpr (MODE == 3) r[i] = c * dppf<0x101, true>(r[i]);                            // v_mul_f32_dpp row_shl:1
            if constexpr (MODE == 4) { float z = r[i]; asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(z) : "v"(s), "v"(c)); r[i] = z; }
            if constexpr (MODE == 5) { const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r[i]), i)); r[i] = r[i] * t; }  // v_readlane + v_mul (SGPR operand)
            if constexpr (MODE == 6) r[i] = (lane == i) ? c : r[i];                                  // v_cmp + v_cndmask (or hoisted mask)
            if constexpr (MODE == 7) { float z = r[i]; asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(z) : "v"(s), "v"(c)); r[i] = z; }
            if constexpr (MODE == 8) { float z = r[i]; asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(z) : "v"(s), "v"(c)); r[i] = z; }
            if constexpr (MODE == 9) { auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[i]), __float_as_uint(r[(i + 1) & 63]), false, false); r[i] = __uint_as_float(sw[0]); }
            if constexpr (MODE == 10) r[i] = __shfl_xor(r[i], 5, 64);                               // ds_bpermute
            if constexpr (MODE == 11) {                                                             // v_pk_fma_f32: two FMAs per instruction (one per even i)
                if ((i & 1) == 0) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 x = {r[i], r[i + 1]}, cs = {c, s}, y;
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(y) : "v"(x), "v"(cs));
                    r[i] = y[0]; r[i + 1] = y[1];
                }
            }
            if constexpr (MODE == 12) {                                                             // a row rotation as two packed instructions per register pair
                if ((i & 1) == 0) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 x = {r[i], r[i + 1]}, cs = {c, s}, t, y;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(t) : "v"(x), "v"(cs));       // (s x0, -s x1)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "=v"(y) : "v"(x), "v"(cs), "v"(t));   // (c x1 + s x0, c x0 - s x1)
                    r[i] = y[0]; r[i + 1] = y[1];
                }
            }
        }
        asm volatile("" : "+v"(c), "+v"(s));
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE>
float run(int blocks, int iters, float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 64>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 64>>>(out, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 8192 * 64 * 4);
    const int iters = 2000;
    const char* names[] = {"v_fma_f32", "mul_dpp quad_perm", "mul_dpp wave_shl:1", "mul_dpp row_shl:1", "fmac_dpp wave_shr:1 (asm)", "readlane+mul", "cmp+cndmask",
                           "fmac_dpp row_shr:1 (asm)", "fmac_dpp quad_perm (asm)", "permlane32_swap", "ds_bpermute (shfl_xor)", "v_pk_fma_f32 (32 instr = 64 FMA)",
                           "row rotation packed (64 instr per 32 pairs)"};
    for (int blocks : {1024, 2048, 4096}) {
        float ms[13];
        ms[0] = run<0>(blocks, iters, out); ms[1] = run<1>(blocks, iters, out); ms[2] = run<2>(blocks, iters, out); ms[3] = run<3>(blocks, iters, out);
        ms[4] = run<4>(blocks, iters, out); ms[5] = run<5>(blocks, iters, out); ms[6] = run<6>(blocks, iters, out); ms[7] = run<7>(blocks, iters, out);
        ms[8] = run<8>(blocks, iters, out); ms[9] = run<9>(blocks, iters, out); ms[10] = run<10>(blocks, iters, out);
        ms[11] = run<11>(blocks, iters, out); ms[12] = run<12>(blocks, iters, out);
        for (int m = 0; m < 13; ++m) {
            // ns per 64-register body per wave-slot: waves per SIMD = blocks / 1024
            const double ns_per_op = 1e6 * ms[m] / (iters * 64.0);
            printf("{\"waves\": %d, \"op\": \"%s\", \"ms\": %.3f, \"ns_per_wave_instr_group\": %.3f, \"ns_per_instr_per_simd\": %.3f}\n", blocks, names[m], ms[m], ns_per_op,
                   ns_per_op / (blocks / 1024.0));
        }
    }
    return 0;
}
