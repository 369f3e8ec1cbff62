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

// ---- cache maintenance at kernel boundaries --------------------------------------------------------
// The sweeps CAN run independent problem groups on several HIP streams at once (ASVD_GROUPS > 1; default 1 since round 2).
// Measured on MI355X / ROCm 7.2: with kernels of OTHER streams in flight, data a kernel leaves in its XCD's L2 (small buffers that
// are re-written every step: carried Gram blocks, 64x64 Q's, flags) was not reliably visible to the next kernel of the SAME stream
// when that ran on another XCD — nondeterministic stale reads (the Jacobi iteration absorbs them as extra sweeps, which is how they
// were found), never with a single stream.  With more than one stream group every kernel of the sweeps therefore starts with an
// agent-scope acquire (invalidate this CU's L1) and ends with an agent-scope release (write back the XCD L2's dirty lines); either
// fence alone was not enough.  The release costs a full L2 write-back per workgroup (the streaming kernels run 2x slower with
// it), which is more than the 5-9 % the overlap of stream groups buys: one stream group, no fences, is the default.
// The flag travels with every launch (Sched::fence, a kernel ARGUMENT): nothing the kernels read is process-global state, so concurrent
// calls with different settings cannot disturb each other.
// fence bits: 1 = acquire at kernel start, 2 = release at kernel end (ASVD_FENCE=1 means both; 2 / 3 = acquire / release only, experiments)
#define ASVD_KERNEL_ACQUIRE(sc) do { if ((sc).fence & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)
#define ASVD_KERNEL_RELEASE(sc) do { if ((sc).fence & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); } while (0)

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
