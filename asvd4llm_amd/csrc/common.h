// common.h — shared device/host helpers for libasvd_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/asvd_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define ASVD_HIP_CHECK(expr)                    \
    do {                                        \
        hipError_t _e = (expr);                 \
        if (_e != hipSuccess) return ASVD_E_HIP; \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up64(int64_t a, int64_t b) { return ceil_div64(a, b) * b; }

// ---- element conversion (device) ------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return __half2float(__ushort_as_half(b)); }
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) { return __half_as_ushort(__float2half_rn(f)); }

template <int DT> struct elem;
template <> struct elem<ASVD_F32> {
    typedef float T;
    static __device__ __forceinline__ float ld(const void* p, int64_t i) { return ((const float*)p)[i]; }
    static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct elem<ASVD_F16> {
    typedef uint16_t T;
    static __device__ __forceinline__ float ld(const void* p, int64_t i) { return f16_bits_to_f32(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = f32_to_f16_bits(v); }
    static __device__ __forceinline__ float rnd(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }
};
template <> struct elem<ASVD_BF16> {
    typedef uint16_t T;
    static __device__ __forceinline__ float ld(const void* p, int64_t i) { return bf16_bits_to_f32(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float rnd(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

// run-time dtype dispatch for kernels templated on one dtype
#define ASVD_DISPATCH_DTYPE(dt, NAME, ...)                              \
    switch (dt) {                                                       \
        case ASVD_F32: { constexpr int NAME = ASVD_F32; __VA_ARGS__; break; }   \
        case ASVD_F16: { constexpr int NAME = ASVD_F16; __VA_ARGS__; break; }   \
        case ASVD_BF16: { constexpr int NAME = ASVD_BF16; __VA_ARGS__; break; } \
        default: return ASVD_E_BADARG;                                  \
    }

static inline size_t dtype_size(int dt) { return dt == ASVD_F32 ? 4 : 2; }
static inline bool dtype_ok(int dt) { return dt == ASVD_F32 || dt == ASVD_F16 || dt == ASVD_BF16; }

// ---- one stream per call --------------------------------------------------------------------------
// Every launch of a call goes to the caller's stream; kernel boundaries on that stream order everything.  (Rounds 1-2 drove independent problem
// groups on several streams; what looked like missing cache maintenance between them was a race inside the LDS eigen-solver of the time — gone
// with that solver in round 3 — and the overlap bought nothing once single launches filled the chip, so the stream groups, their agent-scope
// fences and the solver are all gone: DESIGN.md 3.8.)  Concurrent CALLS on different streams with disjoint workspaces are supported and tested
// (tests/test_gpu_concurrency.py): nothing the kernels read is process-global state, schedules travel by value in the Sched argument.

// ---- small control words rewritten between launches (activity flags, done flags, pair lists) ----------------------------------
// Read through the VECTOR path with an agent-scope load (global_load ... sc1: served by L2, never by the scalar data cache or this
// CU's L1).  A plain `flags[i]` with a wave-uniform index is compiled to s_load_dword, i.e. it goes through the scalar cache that
// several CUs share; these words are rewritten at the same address every step of the sweep.
__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- wave / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
