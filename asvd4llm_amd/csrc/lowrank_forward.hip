// lowrank_forward.hip — K9: fused SVDLinear forward  y = (x Bᵀ) Aᵀ + bias  for small token counts (decode / short prefill).
//
// Replaces the two dependent nn.Linear launches of /root/reference/modules/svd_linear.py:105-109
//     y = self.BLinear(inp); y = self.ALinear(y)
// by ONE persistent launch.  At T <= 128 tokens the op is a pure weight stream (r(K+N) fp16 values, read exactly once) so the design
// goal is: every byte of B and A crosses HBM once in >=128-byte row segments, the r-wide intermediate z never goes to HBM as a
// tensor of its own launch, and there is no second launch / no hipBLASLt heuristics call on the critical path.
//
//   phase 1   unit = (32-token tile, slice of SL ranks):  z[t, r0:r0+SL] = fp16( sum_k x[t,k] B[r0+j,k] )   (fp32 MFMA accumulate,
//             rounded to fp16 exactly where the reference's BLinear output is rounded)
//   grid barrier (all workgroups are co-resident: grid <= number of CUs)
//   phase 2   unit = (32-token tile, tile of SL2 output features):  y[t, n] = fp16( sum_j z[t,j] A[n,j] + bias[n] )
//
// z ([Tpad, rp] fp16, <= 0.5 MB) lives in a caller-provided scratch buffer; it is produced and consumed inside the launch and stays in
// L2 / MALL.  Both phases use v_mfma_f32_32x32x16_f16 with operands loaded STRAIGHT from global memory: the reduction index of an MFMA
// may be permuted freely as long as both operands use the same permutation, so lane (i, g) of a wave loads the 64 contiguous
// bytes  row i, k = 64*it + 32*g .. +31  and feeds 8-value chunk m of them to MFMA m of the iteration — a row is read in
// 128-byte segments with no LDS staging.  The four waves of a workgroup split the k iterations and reduce through LDS (16 KB).
// With few tokens the 32-wide MFMA tile is half-masked (SL = 16) to get twice the workgroups streaming: the op is bandwidth-,
// not MFMA-bound.
//
// Layout contract (the Python module pads once at construction):  Bp [rp, K] = B with zero rows appended, Ap [N, rp] = A with zero
// columns appended, rp a multiple of 64, K a multiple of 64.
#include "common.h"

namespace {

// The intermediate z is the only data that crosses the grid barrier.  It is written with agent-scope (sc1) stores, which go through to
// memory, and read back with agent-scope loads, which do not hit a stale line of another XCD's L2 — so the barrier itself needs NO
// cache-wide fence.  (First version: an agent-scope release + acquire fence in every wave = a write-back / invalidate of the whole L2
// per wave; 1024-2048 of them made the kernel 36-75 us long whatever the data path did.)
__device__ __forceinline__ void z_store(uint16_t* p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint4 z_load16(const uint4* p) {
    const uint64_t* q = (const uint64_t*)p;
    const uint64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
}

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nwg) {
    // sense-reversing barrier on three words: bar[0] arrivals, bar[1] generation, bar[2] give-up flag.  Zero-initialised once by the caller;
    // reusable.  The generation is read BEFORE this workgroup arrives: both words sit in one 128-byte line (one L2 channel serves its
    // requests in issue order) and the compiler barrier keeps the program order.  The spin is BOUNDED: the launch is not cooperative, so
    // co-residency of the <= #CUs workgroups is an assumption (another stream's persistent kernel could hold CUs); after ~2^22 polls a
    // waiter sets bar[2] and leaves instead of hanging the GPU.  What it computes next comes from an incomplete z, so the failure is made
    // VISIBLE: every workgroup ends by overwriting the elements of y it stored with NaN when the flag is set (gave_up).  (A waiter gives
    // up only while the barrier is incomplete, so every workgroup that passes normally does so AFTER the flag was set and sees it at its end.)
    // ops.lowrank_forward additionally reads and clears the word under ASVD_STRICT / ASVD_DEBUG and raises.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's z stores have been acknowledged (s_waitcnt vmcnt(0)); no L2 write-back
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const unsigned prev = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nwg - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);  // the reset is out before the generation moves
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {  // seconds: not a slow barrier, a workgroup that never became resident
                    __hip_atomic_store(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// last statement of both kernels: a launch in which some workgroup left the barrier without its peers has no valid output.  Every workgroup
// overwrites exactly the elements of y IT stored, after its own stores (a slice of all of y would race with the stores of workgroups that
// are still in phase 2: NaN mixed with finite garbage).  The flag is read after the workgroup's last store: a workgroup can only give up
// while the barrier is incomplete, i.e. before any workgroup starts phase 2.
__device__ __forceinline__ bool gave_up(const unsigned* bar) { return __hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }

// One 32 x 32 output tile  C[i][j] = sum_k P[prow0+i][k] * Q[qrow0+j][k]  over k in [0, 64*nk), fp32, all four waves.
// Rows i >= pvalid of P and j >= qvalid of Q read as zero.  Result is left in red[] (sum of the 4 wave partials is done by the caller).
template <bool PCOH>  // PCOH: P is the intermediate z — agent-scope loads (see z_load16)
__device__ __forceinline__ void tile_partial(f32x16& acc, const uint16_t* __restrict__ P, int64_t ldp, int pvalid,
                                             const uint16_t* __restrict__ Q, int64_t ldq, int qvalid, int nk, int wave, int lane) {
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // masked rows load a valid (clamped) row and are zeroed in registers: no divergent loads
    const unsigned pm = i < pvalid ? 0xffffffffu : 0u, qm = i < qvalid ? 0xffffffffu : 0u;
    const uint4* prow = (const uint4*)(P + (int64_t)min(i, pvalid - 1) * ldp + 32 * g);
    const uint4* qrow = (const uint4*)(Q + (int64_t)min(i, qvalid - 1) * ldq + 32 * g);
    // One workgroup per CU means one wave per SIMD: latency is hidden only by loads in flight, so each wave issues the loads of FOUR
    // of its iterations (32 x 16 bytes per lane) before the first MFMA.  A group's tail iterations re-load a valid one and are masked.
    for (int base = wave; base < nk; base += 16) {
        uint4 pv[4][4], qv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = (base + 4 * u < nk) ? base + 4 * u : base;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                pv[u][m] = PCOH ? z_load16(prow + it * 8 + m) : prow[it * 8 + m];  // 64 halfs = 8 uint4 per iteration per row; this lane's 4 start at 4*g (folded into prow)
                qv[u][m] = qrow[it * 8 + m];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned live = (base + 4 * u < nk) ? 0xffffffffu : 0u;
            const unsigned pmu = pm & live;  // masking one operand is enough to drop the product
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint4 pw = pv[u][m], qw = qv[u][m];
                pw.x &= pmu, pw.y &= pmu, pw.z &= pmu, pw.w &= pmu;
                qw.x &= qm, qw.y &= qm, qw.z &= qm, qw.w &= qm;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pw), __builtin_bit_cast(f16x8, qw), acc, 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256, 1) void lowrank_forward_kernel(const uint16_t* __restrict__ x, int T, const uint16_t* __restrict__ Bp,
                                                                 const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ bias, int N,
                                                                 int K, int rp, uint16_t* __restrict__ y, uint16_t* __restrict__ z,
                                                                 unsigned* bar, int sl1, int sl2) {
    __shared__ float red[4][16 * 64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nT = (T + 31) / 32;
    const int G = gridDim.x;

    // ---- phase 1: z = fp16(x Bᵀ) ---------------------------------------------------------------------------------------------
    const int ns1 = rp / sl1;
    for (int u = blockIdx.x; u < nT * ns1; u += G) {
        const int t0 = (u / ns1) * 32, r0 = (u % ns1) * sl1;
        f32x16 acc;
        tile_partial<false>(acc, x + (int64_t)t0 * K, K, T - t0, Bp + (int64_t)r0 * K, K, sl1, K / 64, wave, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave][q * 64 + lane] = acc[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, reg = e >> 6, l = e & 63;
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), col = l & 31;
            const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            if (col < sl1) z_store(z + (int64_t)(t0 + row) * rp + r0 + col, f32_to_f16_bits(v));  // z has 32*nT rows: no token mask needed
        }
        __syncthreads();
    }

    grid_barrier(bar, (unsigned)G);

    // ---- phase 2: y = fp16(z Aᵀ + bias) --------------------------------------------------------------------------------------
    const int ns2 = (N + sl2 - 1) / sl2;
    for (int u = blockIdx.x; u < nT * ns2; u += G) {
        const int t0 = (u / ns2) * 32, n0 = (u % ns2) * sl2;
        const int nvalid = min(sl2, N - n0);
        f32x16 acc;
        tile_partial<true>(acc, z + (int64_t)t0 * rp, rp, 32, Ap + (int64_t)n0 * rp, rp, nvalid, rp / 64, wave, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wave][q * 64 + lane] = acc[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, reg = e >> 6, l = e & 63;
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), col = l & 31;
            if (col < nvalid && t0 + row < T) {
                float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
                if (bias) v += f16_bits_to_f32(bias[n0 + col]);
                y[(int64_t)(t0 + row) * N + n0 + col] = f32_to_f16_bits(v);
            }
        }
        __syncthreads();
    }
    if (gave_up(bar))   // poison this workgroup's own tiles
        for (int u = blockIdx.x; u < nT * ns2; u += G) {
            const int t0 = (u / ns2) * 32, n0 = (u % ns2) * sl2, nvalid = min(sl2, N - n0);
            for (int e = tid; e < 1024; e += 256) {
                const int row = e >> 5, col = e & 31;
                if (col < nvalid && t0 + row < T) y[(int64_t)(t0 + row) * N + n0 + col] = 0x7e00;  // fp16 NaN
            }
        }
}

// --------------------------------------------------------------------------------------------------
// Decode-sized inputs (T <= 4 tokens): the same two phases as a pair of fused GEMVs.  A wave takes one weight ROW at a time and reads it
// with fully coalesced 16-byte loads (1 KiB per wave-instruction, the whole row in flight at once); the T activation rows sit in LDS
// (phase 1: x, phase 2: z); products are v_dot2_f32_f16 (exact fp16 products, fp32 accumulation), the 64 partial sums of a row are
// folded with DPP/shuffle steps.  Eight waves per workgroup, one workgroup per CU (the grid barrier needs co-residency), rows
// interleaved over all waves of the grid.  No MFMA: at <= 4 tokens the tile would be >= 87 % padding and the op is a pure weight stream.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int GV_MAXC = 22;  // chunks of 64 uint4 (512 halfs) per weight row: K, rp <= 11264

// all 16-byte pieces of weight row `row` (klen8 of them) into registers: the whole row in flight at once
__device__ __forceinline__ void gemv_load(uint4 (&wv)[GV_MAXC], const uint4* __restrict__ row, int klen8, int lane) {
    const int nch = (klen8 + 63) / 64;
#pragma unroll
    for (int c = 0; c < GV_MAXC; ++c)
        if (c < nch) {
            const int k8 = c * 64 + lane;
            wv[c] = k8 < klen8 ? row[k8] : make_uint4(0, 0, 0, 0);
        }
}

// rows gw, gw + nw, ... of W against the TT activation rows in LDS.  The FIRST row of the wave arrives already loaded in wv: the caller
// issues it before it waits for the activations (x -> LDS, or the grid barrier and z -> LDS), so that wait overlaps the weight stream.
template <int TT, bool ZOUT>
__device__ __forceinline__ void gemv_rows(uint4 (&wv)[GV_MAXC], const uint4* __restrict__ W4, int64_t ldw4 /* row stride in uint4 */, int rows,
                                          int klen8 /* k length in uint4 */, const uint4* __restrict__ xs /* LDS: [TT][klen8] */, int gw, int nw,
                                          int lane, uint16_t* __restrict__ out, int64_t ldo, const uint16_t* __restrict__ bias, int T) {
    const int nch = (klen8 + 63) / 64;
    for (int j = gw; j < rows; j += nw) {
        if (j != gw) gemv_load(wv, W4 + (int64_t)j * ldw4, klen8, lane);
        float acc[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = 0.f;
#pragma unroll
        for (int c = 0; c < GV_MAXC; ++c)
            if (c < nch) {
                const int k8 = min(c * 64 + lane, klen8 - 1);  // masked lanes hold zeros in wv: any valid x slot will do
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const uint4 xv = xs[t * klen8 + k8];
                    acc[t] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv[c].x), __builtin_bit_cast(f16x2, xv.x), acc[t], false);
                    acc[t] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv[c].y), __builtin_bit_cast(f16x2, xv.y), acc[t], false);
                    acc[t] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv[c].z), __builtin_bit_cast(f16x2, xv.z), acc[t], false);
                    acc[t] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, wv[c].w), __builtin_bit_cast(f16x2, xv.w), acc[t], false);
                }
            }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float v = acc[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && t < T) {
                if (bias) v += f16_bits_to_f32(bias[j]);
                if (ZOUT) z_store(out + (int64_t)t * ldo + j, f32_to_f16_bits(v));
                else out[(int64_t)t * ldo + j] = f32_to_f16_bits(v);
            }
        }
    }
}

template <int TT>
__global__ __launch_bounds__(512, 1) void lowrank_gemv_kernel(const uint16_t* __restrict__ x, int T, const uint16_t* __restrict__ Bp,
                                                              const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ bias, int N, int K,
                                                              int rp, uint16_t* __restrict__ y, uint16_t* __restrict__ z, unsigned* bar) {
    extern __shared__ __attribute__((aligned(16))) uint4 gv_smem[];  // [TT][max(K, rp) / 8]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int G = gridDim.x, gw = blockIdx.x * 8 + wave, nw = G * 8;
    uint4 wv[GV_MAXC];
    // ---- phase 1: z[t, j] = fp16( x[t, :] . B[j, :] ), j < rp (padded ranks have zero rows in Bp and give exact zeros) ----
    const int k8 = K / 8, r8 = rp / 8;
    if (gw < rp) gemv_load(wv, (const uint4*)Bp + (int64_t)gw * k8, k8, lane);  // weight row first, activations second
    for (int e = tid; e < TT * k8; e += 512) {
        const int t = e / k8, kk = e - t * k8;
        gv_smem[e] = t < T ? ((const uint4*)x)[(int64_t)t * k8 + kk] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    gemv_rows<TT, true>(wv, (const uint4*)Bp, k8, rp, k8, gv_smem, gw, nw, lane, z, rp, nullptr, TT);

    // the first A row of this wave does not depend on phase 1: it streams in while the grid waits at the barrier
    if (gw < N) gemv_load(wv, (const uint4*)Ap + (int64_t)gw * r8, r8, lane);
    grid_barrier(bar, (unsigned)G);

    // ---- phase 2: y[t, n] = fp16( z[t, :] . A[n, :] + bias[n] ) ----
    for (int e = tid; e < TT * r8; e += 512) gv_smem[e] = z_load16((const uint4*)z + e);
    __syncthreads();
    gemv_rows<TT, false>(wv, (const uint4*)Ap, r8, N, r8, gv_smem, gw, nw, lane, y, N, bias, T);
    if (gave_up(bar) && lane == 0)   // poison the rows this wave stored (gemv_rows: rows gw, gw + nw, ...)
        for (int j = gw; j < N; j += nw)
            for (int t = 0; t < T; ++t) y[(int64_t)t * N + j] = 0x7e00;  // fp16 NaN
}

}  // namespace

extern "C" {

int64_t asvd_lowrank_padded_rank(int64_t r) { return round_up64(r, 64); }

size_t asvd_lowrank_work_bytes(int64_t T, int64_t rp) {
    // [barrier: 256 bytes][z: 32*ceil(T/32) x rp fp16]
    return 256 + (size_t)(round_up64(T, 32) * rp) * sizeof(uint16_t);
}

int asvd_lowrank_forward_f16(const void* x, int64_t T, const void* Bp, const void* Ap, const void* bias, int64_t N, int64_t K, int64_t rp,
                             void* y, void* work, size_t work_bytes, void* stream) {
    if (!x || !Bp || !Ap || !y || !work) return ASVD_E_BADARG;
    if (T < 1 || T > ASVD_LOWRANK_MAX_TOKENS || N < 1 || K < 64 || (K % 64) || rp < 64 || (rp % 64)) return ASVD_E_BADARG;
    if (work_bytes < asvd_lowrank_work_bytes(T, rp)) return ASVD_E_WORKSPACE;
    if ((((uintptr_t)x) | ((uintptr_t)Bp) | ((uintptr_t)Ap) | ((uintptr_t)work)) & 15) return ASVD_E_BADARG;
    // per-call host cost matters at decode size: the CU count and the LDS opt-in are looked up once per device / kernel
    static int s_cus[16] = {0};
    static bool s_attr[16][3] = {{false}};
    int dev = 0;
    ASVD_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return ASVD_E_BADARG;
    if (!s_cus[dev]) ASVD_HIP_CHECK(hipDeviceGetAttribute(&s_cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    const int cus = s_cus[dev];
    const int nT = (int)((T + 31) / 32);
    // half-masked tiles when full ones would leave more than half of the CUs without a unit (bandwidth-bound: more streams win)
    const int sl1 = (nT * (rp / 32) >= cus / 2) ? 32 : 16;
    const int sl2 = (nT * ((N + 31) / 32) >= cus / 2) ? 32 : 16;
    const int64_t units = (int64_t)nT * ((rp / sl1) > ((N + sl2 - 1) / sl2) ? (rp / sl1) : ((N + sl2 - 1) / sl2));
    const int grid = (int)(units < cus ? units : cus);  // <= one workgroup per CU: co-resident, the in-kernel barrier cannot deadlock
    unsigned* bar = (unsigned*)work;
    uint16_t* z = (uint16_t*)((char*)work + 256);
    const size_t gv_lds = (size_t)(T <= 1 ? 1 : (T <= 2 ? 2 : 4)) * (size_t)(K > rp ? K : rp) * sizeof(uint16_t);
    if (T <= 4 && K <= 11264 && rp <= 11264 && gv_lds <= 96 * 1024) {  // decode-sized: fused GEMV pair (see lowrank_gemv_kernel)
        const int64_t rows = rp > N ? rp : N;
        const int ggrid = (int)(ceil_div64(rows, 8) < cus ? ceil_div64(rows, 8) : cus);
#define ASVD_GV_LAUNCH(TT)                                                                                                                    \
    do {                                                                                                                                      \
        bool& attr_done = s_attr[dev][TT == 1 ? 0 : (TT == 2 ? 1 : 2)];                                                                         \
        if (!attr_done) {                                                                                                                     \
            ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)lowrank_gemv_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));   \
            attr_done = true;                                                                                                                 \
        }                                                                                                                                     \
        hipLaunchKernelGGL(lowrank_gemv_kernel<TT>, dim3(ggrid), dim3(512), gv_lds, (hipStream_t)stream, (const uint16_t*)x, (int)T,            \
                           (const uint16_t*)Bp, (const uint16_t*)Ap, (const uint16_t*)bias, (int)N, (int)K, (int)rp, (uint16_t*)y, z, bar);     \
    } while (0)
        if (T <= 1) ASVD_GV_LAUNCH(1);
        else if (T <= 2) ASVD_GV_LAUNCH(2);
        else ASVD_GV_LAUNCH(4);
#undef ASVD_GV_LAUNCH
        ASVD_HIP_CHECK(hipGetLastError());
        return ASVD_OK;
    }
    hipLaunchKernelGGL(lowrank_forward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (int)T,
                       (const uint16_t*)Bp, (const uint16_t*)Ap, (const uint16_t*)bias, (int)N, (int)K, (int)rp, (uint16_t*)y, z, bar, sl1,
                       sl2);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

}  // extern "C"
