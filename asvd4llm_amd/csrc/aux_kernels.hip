// aux_kernels.hip — the HBM-bound kernels around the SVD: calibration-hook statistics (K1/K2), scale vector and
// column scaling (K3), truncate/un-scale/fuse/split (K5/K6), Frobenius norm (K8) and the reconstruction-error
// evidence kernel (K9).  Reference lines are cited at each entry point in include/asvd_hip.h.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

// --------------------------------------------------------------------------------------------------
// K1/K2  |x| column statistics.  A wave covers 64*VEC consecutive columns of one row with 16-B loads per lane
// (1 KiB per wave-instruction); the 4 waves of a workgroup interleave rows; row range split over blockIdx.y.
template <int DT> struct vec_of;  // VEC elements per 16-byte load
template <> struct vec_of<ASVD_F32> { static constexpr int VEC = 4; };
template <> struct vec_of<ASVD_F16> { static constexpr int VEC = 8; };
template <> struct vec_of<ASVD_BF16> { static constexpr int VEC = 8; };

template <int DT, int MODE>
__global__ __launch_bounds__(256) void absstat_partial_kernel(const void* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                              int64_t rows_per_split, float* __restrict__ part,
                                                              int* __restrict__ nanflag) {
    constexpr int VEC = vec_of<DT>::VEC;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * VEC;
    const int split = blockIdx.y;
    const int64_t rb = split * rows_per_split;
    const int64_t re = (rb + rows_per_split < rows) ? rb + rows_per_split : rows;
    float acc[VEC];
    int seen_nan = 0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
    const bool full = (c0 + VEC <= cols) && ((ld % VEC) == 0) && ((((uintptr_t)x) & 15) == 0);
    auto consume = [&](const float* vals) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float a = (MODE == ASVD_STAT_SQ_MEAN) ? elem<DT>::rnd(vals[v] * vals[v]) : fabsf(vals[v]);
            if (MODE != ASVD_STAT_ABS_MAX) acc[v] += a;
            else {
                if (a != a) seen_nan |= (1 << v);
                else acc[v] = fmaxf(acc[v], a);
            }
        }
    };
    auto unpack16 = [&](const uint4& t, float* vals) {
        const uint32_t wds[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const uint16_t bits = (uint16_t)(wds[(v >> 1) & 3] >> (16 * (v & 1)));
            vals[v] = (DT == ASVD_F16) ? f16_bits_to_f32(bits) : bf16_bits_to_f32(bits);
        }
    };
    int64_t r = rb + w;
    if (full) {
        // four rows of this wave per iteration, all four 16-byte loads issued before the first use: a wave only has rows/(4 splits) rows
        // to read, so the kernel's length is the number of DEPENDENT load latencies, not the bytes (rows are summed in the same order)
        for (; r + 12 < re; r += 16) {
            float vals[VEC];
            if (DT == ASVD_F32) {
                f32x4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *(const f32x4*)((const float*)x + (r + 4 * u) * ld + c0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) vals[v] = t[u][v % 4];
                    consume(vals);
                }
            } else {
                uint4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *(const uint4*)((const uint16_t*)x + (r + 4 * u) * ld + c0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unpack16(t[u], vals);
                    consume(vals);
                }
            }
        }
    }
    for (; r < re; r += 4) {
        float vals[VEC];
        if (full) {
            if (DT == ASVD_F32) {
                const f32x4 t = *(const f32x4*)((const float*)x + r * ld + c0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) vals[v] = t[v % 4];
            } else {
                const uint4 t = *(const uint4*)((const uint16_t*)x + r * ld + c0);
                unpack16(t, vals);
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) vals[v] = (c0 + v < cols) ? elem<DT>::ld(x, r * ld + c0 + v) : 0.0f;
        }
        consume(vals);
    }
    __shared__ float red[4][64 * VEC];
    __shared__ int rnan[4][64];
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[w][v * 64 + lane] = acc[v];
    rnan[w][lane] = seen_nan;
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float s = red[0][v * 64 + lane];
            if (MODE != ASVD_STAT_ABS_MAX) s = ((s + red[1][v * 64 + lane]) + red[2][v * 64 + lane]) + red[3][v * 64 + lane];
            else s = fmaxf(fmaxf(s, red[1][v * 64 + lane]), fmaxf(red[2][v * 64 + lane], red[3][v * 64 + lane]));
            if (c0 + v < cols) part[(int64_t)split * cols + c0 + v] = s;
        }
        if (MODE == ASVD_STAT_ABS_MAX) {
            const int nn = rnan[0][lane] | rnan[1][lane] | rnan[2][lane] | rnan[3][lane];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (c0 + v < cols) nanflag[(int64_t)split * cols + c0 + v] = (nn >> v) & 1;
        }
    }
}

// 64 columns per workgroup; the four waves each fold every fourth partial (four independent loads in flight), then wave 0 combines
// the four in a fixed order: a dependent chain of nsplit/4 instead of nsplit loads (the finalize used to cost as much as the pass over X)
template <int AT, int MODE>
__global__ __launch_bounds__(256) void absstat_final_kernel(const float* __restrict__ part, const int* __restrict__ nanflag, int nsplit, int64_t rows,
                                                            int64_t cols, void* __restrict__ acc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = c < cols;
    float s = 0.0f;
    int nn = 0;
    if (ok) {
        int k = w;
        for (; k + 12 < nsplit; k += 16) {
            const float v0 = part[(int64_t)k * cols + c], v1 = part[(int64_t)(k + 4) * cols + c], v2 = part[(int64_t)(k + 8) * cols + c],
                        v3 = part[(int64_t)(k + 12) * cols + c];
            if (MODE != ASVD_STAT_ABS_MAX) s = (((s + v0) + v1) + v2) + v3;
            else {
                s = fmaxf(fmaxf(s, v0), fmaxf(v1, fmaxf(v2, v3)));
                nn |= nanflag[(int64_t)k * cols + c] | nanflag[(int64_t)(k + 4) * cols + c] | nanflag[(int64_t)(k + 8) * cols + c] |
                      nanflag[(int64_t)(k + 12) * cols + c];
            }
        }
        for (; k < nsplit; k += 4) {
            const float v = part[(int64_t)k * cols + c];
            if (MODE != ASVD_STAT_ABS_MAX) s += v;
            else { s = fmaxf(s, v); nn |= nanflag[(int64_t)k * cols + c]; }
        }
    }
    __shared__ float red[4][64];
    __shared__ int rnan[4][64];
    red[w][lane] = s;
    rnan[w][lane] = nn;
    __syncthreads();
    if (w != 0 || !ok) return;
    const float old = elem<AT>::ld(acc, c);
    if (MODE != ASVD_STAT_ABS_MAX) {
        const float tot = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const float mean = elem<AT>::rnd(__fdiv_rn(tot, (float)rows));  // .mean() result in the activation dtype
        elem<AT>::st(acc, c, old + mean);                               // `+=` in that dtype (one rounding)
    } else {
        const float mx = fmaxf(fmaxf(red[0][lane], red[1][lane]), fmaxf(red[2][lane], red[3][lane]));
        const int anynan = rnan[0][lane] | rnan[1][lane] | rnan[2][lane] | rnan[3][lane];
        // torch.where(abs_max > acc, abs_max, acc): a NaN abs_max never wins
        if (!anynan && mx > old) elem<AT>::st(acc, c, mx);
    }
}

// --------------------------------------------------------------------------------------------------
// K3a  s = (scaling**alpha [* fisher**alpha]) + eps, each op rounded to the statistics dtype
__device__ __forceinline__ float pow_alpha(float x, float alpha) {
    if (alpha == 0.5f) return (float)sqrt((double)x);  // correctly rounded (fp64 sqrt rounded once; v_sqrt_f32 is 1 ulp)
    if (alpha == 1.0f) return x;
    if (alpha == 2.0f) return x * x;
    return (float)pow((double)x, (double)alpha);  // double pow rounded once: within the last fp32 bit of a correctly rounded powf
}
template <int DT>
__global__ void make_scale_kernel(const void* __restrict__ scaling, const void* __restrict__ fisher, int64_t n, float alpha,
                                  float eps, void* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // torch casts the python-scalar exponent to the tensor dtype before pow (0.3 -> 0.30005 for fp16): do the same
    alpha = elem<DT>::rnd(alpha);
    float p = elem<DT>::rnd(pow_alpha(elem<DT>::ld(scaling, i), alpha));
    if (fisher) {
        const float f = elem<DT>::rnd(pow_alpha(elem<DT>::ld(fisher, i), alpha));
        p = elem<DT>::rnd(p * f);
    }
    elem<DT>::st(out, i, p + eps);
}

// K3b  out = float(w) * float(s)
template <int DT, int ST>
__global__ __launch_bounds__(256) void scale_cols_kernel(const void* __restrict__ w, int64_t m, int64_t n, int64_t ldw,
                                                         const void* __restrict__ s, int has_scale, float* __restrict__ out,
                                                         int64_t ldo) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float sc = has_scale ? elem<ST>::ld(s, j) : 1.0f;
    const int64_t r0 = (int64_t)blockIdx.y * 32;
    const int64_t r1 = (r0 + 32 < m) ? r0 + 32 : m;
    for (int64_t r = r0; r < r1; ++r) out[r * ldo + j] = elem<DT>::ld(w, r * ldw + j) * sc;
}

// --------------------------------------------------------------------------------------------------
// K5/K6  A = U[:, :r] * f(S) ; B = ((V[:, :r] / s[:,None]).T) * g(S)[:,None] ; NaN flags
template <int OT>
__global__ __launch_bounds__(256) void split_a_kernel(const float* __restrict__ U, int64_t ldu, const float* __restrict__ S,
                                                      int64_t m, int64_t r, int fuse, void* __restrict__ A,
                                                      int* __restrict__ nan_flags) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= r) return;
    const float sv = S[j];
    const float f = (fuse == ASVD_FUSE_UV) ? (float)sqrt((double)sv) : (fuse == ASVD_FUSE_U ? sv : 1.0f);
    int bad_u = 0;
    const int64_t i0 = (int64_t)blockIdx.y * 32;
    const int64_t i1 = (i0 + 32 < m) ? i0 + 32 : m;
    for (int64_t i = i0; i < i1; ++i) {
        const float u = U[i * ldu + j];
        bad_u |= (u != u);
        elem<OT>::st(A, i * r + j, (fuse == ASVD_FUSE_V) ? u : u * f);
    }
    if (nan_flags) {
        if (blockIdx.y == 0 && sv != sv) atomicOr(&nan_flags[0], 1);
        if (bad_u) atomicOr(&nan_flags[1], 1);
    }
}

template <int OT, int ST>
__global__ __launch_bounds__(256) void split_b_kernel(const float* __restrict__ V, int64_t ldv, const float* __restrict__ S,
                                                      const void* __restrict__ s, int has_scale, int64_t n, int64_t r,
                                                      int fuse, void* __restrict__ B, int* __restrict__ nan_flags) {
    // 32 (i: rows of V) x 32 (j: rank index) tile, transposed through LDS: B[j][i]
    __shared__ float tile[32][33];
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;
    int bad = 0;
    for (int a = ly; a < 32; a += 8) {
        const int64_t i = i0 + a, j = j0 + lx;
        float v = 0.0f;
        if (i < n && j < r) {
            v = V[i * ldv + j];
            if (has_scale) v = __fdiv_rn(v, elem<ST>::ld(s, i));
            bad |= (v != v);
        }
        tile[a][lx] = v;
    }
    __syncthreads();
    for (int a = ly; a < 32; a += 8) {
        const int64_t j = j0 + a, i = i0 + lx;
        if (i < n && j < r) {
            const float sv = S[j];
            const float g = (fuse == ASVD_FUSE_UV) ? (float)sqrt((double)sv) : (fuse == ASVD_FUSE_V ? sv : 1.0f);
            const float v = tile[lx][a];
            elem<OT>::st(B, j * n + i, (fuse == ASVD_FUSE_U) ? v : v * g);
        }
    }
    if (nan_flags && bad) atomicOr(&nan_flags[2], 1);
}

// --------------------------------------------------------------------------------------------------
// K8  sum of squares, fp32, fixed order: per-block partials then a single-thread-block ordered sum
template <int DT>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const void* __restrict__ w, int64_t m, int64_t n, int64_t ldw,
                                                            float* __restrict__ part) {
    // block handles 8 rows; thread strides over columns
    const int64_t r0 = (int64_t)blockIdx.x * 8;
    float acc = 0.0f;
    for (int64_t r = r0; r < r0 + 8 && r < m; ++r)
        for (int64_t j = threadIdx.x; j < n; j += 256) {
            const float v = elem<DT>::ld(w, r * ldw + j);
            acc += v * v;
        }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void ordered_sum_f32_kernel(const float* __restrict__ part, int64_t np, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < np; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// --------------------------------------------------------------------------------------------------
// K9  |W - A*B|_F^2 and |W|_F^2.  One wave per 32x32 output tile, fp32 MFMA (exact products of the stored factors).
template <int WT, int ABT>
__global__ __launch_bounds__(256) void recon_err_kernel(const void* __restrict__ W, int64_t ldw, const void* __restrict__ A,
                                                        const void* __restrict__ B, int64_t m, int64_t n, int64_t r,
                                                        double* __restrict__ part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int64_t r0 = (int64_t)blockIdx.y * 64 + (w >> 1) * 32;
    const int64_t c0 = (int64_t)blockIdx.x * 64 + (w & 1) * 32;
    f32x16 acc = {0};
    const int64_t row = r0 + c;  // A-operand row of this lane
    const int64_t col = c0 + c;  // B-operand column of this lane
    for (int64_t k0 = 0; k0 < r; k0 += 16) {
        float a[8], b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int64_t k = k0 + h * 8 + t;
            a[t] = (row < m && k < r) ? elem<ABT>::ld(A, row * r + k) : 0.0f;
            b[t] = (col < n && k < r) ? elem<ABT>::ld(B, k * n + col) : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    double e2 = 0.0, w2 = 0.0;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t i = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int64_t j = c0 + c;
        if (i < m && j < n) {
            const float wv = elem<WT>::ld(W, i * ldw + j);
            const double d = (double)wv - (double)acc[reg];
            e2 += d * d;
            w2 += (double)wv * (double)wv;
        }
    }
    e2 = wave_reduce_sum_d(e2);
    w2 = wave_reduce_sum_d(w2);
    __shared__ double red[4][2];
    if (lane == 0) { red[w][0] = e2; red[w][1] = w2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        part[2 * bid + 0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        part[2 * bid + 1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
}
// K9 for 16-bit factors (what SVDLinear holds): tiled NT GEMM on the fp16 / bf16 matrix pipe — products of 16-bit factors are exact there, the
// accumulation is fp32 — fused with the squared-difference reduction against W.  Both operands are first brought to a K-contiguous, zero-padded
// form in the workspace (Ap [m][rp], Bt [n][rp], rp = r rounded up to 64: rows 128-byte aligned, no tails), then a 256-thread workgroup owns a
// 128 x 128 tile of A B (wave (wr, wc): 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator registers) and reads its operands STRAIGHT from global memory
// in MFMA operand order: lane (i, h) loads the 64 contiguous bytes k = 64 it + 32 h .. + 31 of its row and feeds 8-value chunk q to MFMA q (the
// reduction index of an MFMA may be permuted as long as both operands agree), so every row is read as whole 128-byte lines; the loads of chunk
// it + 1 are issued before the 16 MFMAs of chunk it.  The factors are re-read by the workgroups of a tile row / column through L2.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pad_rows16_kernel(const uint16_t* __restrict__ A, int64_t m, int64_t r, int64_t rp, uint16_t* __restrict__ Ap) {
    // rows on gridDim.x (no 65535 limit: a 128256-row lm_head), the columns of a row strided over the workgroup
    const int64_t row = blockIdx.x;
    for (int64_t k = threadIdx.x; k < rp; k += 256) Ap[row * rp + k] = k < r ? A[row * r + k] : (uint16_t)0;
}
// Bt[j][k] = B[k][j] through 32 x 32 tiles (both sides coalesced), zero for k >= r.  grid (ceil(n/32), rp/32)
__global__ __launch_bounds__(256) void transpose_pad16_kernel(const uint16_t* __restrict__ B, int64_t r, int64_t n, int64_t rp, uint16_t* __restrict__ Bt) {
    __shared__ uint16_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t j0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + ty + 8 * u, j = j0 + tx;
        tile[ty + 8 * u][tx] = (k < r && j < n) ? B[k * n + j] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = j0 + ty + 8 * u, k = k0 + tx;
        if (j < n) Bt[j * rp + k] = tile[tx][ty + 8 * u];
    }
}

template <int WT, int BF>   // BF: 1 bf16 factors, 0 fp16
__global__ __launch_bounds__(256, 2) void recon_err16_kernel(const void* __restrict__ W, int64_t ldw, const uint16_t* __restrict__ Ap,
                                                             const uint16_t* __restrict__ Bt, int64_t m, int64_t n, int64_t rp, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
    const int h = lane >> 5, c = lane & 31;
    const int64_t r0 = (int64_t)blockIdx.y * 128 + wr * 64, c0 = (int64_t)blockIdx.x * 128 + wc * 64;
    const uint16_t* pa[2];
    const uint16_t* pb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        pa[t] = Ap + min(r0 + 32 * t + c, m - 1) * rp + 32 * h;   // rows beyond the matrix re-read its last row: masked in the epilogue
        pb[t] = Bt + min(c0 + 32 * t + c, n - 1) * rp + 32 * h;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    u32x4 a[2][4], b[2][4], an[2][4], bn[2][4];
    auto fetch = [&](int64_t it, u32x4 (&ra)[2][4], u32x4 (&rb)[2][4]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[t][q] = *(const u32x4*)(pa[t] + it * 64 + 8 * q);
                rb[t][q] = *(const u32x4*)(pb[t] + it * 64 + 8 * q);
            }
    };
    const int64_t nit = rp / 64;
    fetch(0, a, b);
    for (int64_t it = 0; it < nit; ++it) {
        if (it + 1 < nit) fetch(it + 1, an, bn);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a[i][q]), __builtin_bit_cast(b16x8, b[j][q]), acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a[i][q]), __builtin_bit_cast(h16x8, b[j][q]), acc[i][j], 0, 0, 0);
                }
        if (it + 1 < nit) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[t][q] = an[t][q]; b[t][q] = bn[t][q]; }
        }
    }
    double e2 = 0.0, w2 = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t row = r0 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h, col = c0 + 32 * j + c;
                if (row < m && col < n) {
                    const float wv = elem<WT>::ld(W, row * ldw + col);
                    const double d = (double)wv - (double)acc[i][j][reg];
                    e2 += d * d;
                    w2 += (double)wv * (double)wv;
                }
            }
    e2 = wave_reduce_sum_d(e2);
    w2 = wave_reduce_sum_d(w2);
    __shared__ double red[4][2];
    if (lane == 0) { red[w][0] = e2; red[w][1] = w2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        part[2 * bid + 0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        part[2 * bid + 1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
}
// K9, 16-bit factors, ONE GEMM launch (round 6): the same 128 x 128 tile and wave layout as recon_err16_kernel, but the operands are staged through
// LDS straight from the factors AS SVDLinear HOLDS THEM — A [m][r] (K-contiguous, rows at ANY 2-byte alignment: r = 1843, 2686, ...) and B [r][n]
// (K is the slow index) — so the pad and transpose launches and their (m + n) x rp x 2 bytes of workspace traffic are gone.
//   A tile [128 rows][64 k]: a half-wave reads the 32 aligned dwords that hold one row's 64 values (one coalesced 128-byte segment per load;
//     thread (t >> 5, t & 31) owns dword t & 31 of the rows (t >> 5) + 8 j); a row that starts on an odd element (r odd) is shifted by one half-word
//     with v_alignbyte against the following dword (a second, overlapping load); values with k >= r or row >= m are zeroed.
//   B tile [64 k][128 cols]: 16 lanes read the 256 contiguous bytes of one k row (16-byte loads: needs n % 8 == 0 and a 16-byte aligned pointer —
//     the entry point takes the three-launch path otherwise); a thread holds rows (2 p, 2 p + 1) of its 8 columns, packs (k, k + 1) per column into
//     one 32-bit word with v_perm and writes it to the K-CONTIGUOUS image [col][k]: the transpose costs 16 ds_write_b32 per thread and chunk (their
//     bank conflicts — the row stride must keep ds_read_b128 aligned — cost LDS cycles the matrix pipe does not wait for).
//   Both images have a row stride of 72 half-words (144 B): the row-per-lane ds_read_b128 of a 16-lane group falls on distinct banks.  Two image sets (2 x 36 KB): the loads of chunk it + 1 are issued before the 16 MFMAs of chunk it and written
//   to the other set after them — one barrier per chunk.
constexpr int R16_LD = 72;                       // half-words per image row
constexpr int R16_IMG = 128 * R16_LD;            // half-words per image
template <int WT, int BF>
__global__ __launch_bounds__(256, 2) void recon_err16_fused_kernel(const void* __restrict__ W, int64_t ldw, const uint16_t* __restrict__ A,
                                                                   const uint16_t* __restrict__ B, int64_t m, int64_t n, int64_t r, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) uint16_t img[2][2][R16_IMG];   // [set][A | B]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int h = lane >> 5, c = lane & 31;
    const int64_t R0 = (int64_t)blockIdx.y * 128, C0 = (int64_t)blockIdx.x * 128;
    // ---- A fetch geometry: the 32 lanes of a half-wave read ONE row's 32 dwords (a whole 128-byte segment per load instruction); thread
    // (rg = tid >> 5, l = tid & 31) owns dword l of rows rg + 8 j, j = 0..15.  A row that starts on an odd element needs the following dword too
    // (second, overlapping load: L1 hits) — only when r is odd ----
    const int arg = tid >> 5, al = tid & 31;
    const uint32_t* __restrict__ A32 = (const uint32_t*)A;
    const int64_t last_dw = (m * r - 1) >> 1;
    const bool r_odd = (r & 1) != 0;
    // ---- B fetch geometry: 16 lanes read 256 contiguous bytes (128 columns) of one k row; thread (p4 = tid >> 4, cl = tid & 15) owns the 8 columns
    // 8 cl .. + 7 of the row pairs (2 p, 2 p + 1), p = p4 + 16 jj ----
    const int bp4 = tid >> 4, bcl = tid & 15;
    const int64_t bcol = C0 + 8 * bcl;
    const bool bcol_ok = bcol < n;
    // (Measured, round 6: 104 us per launch at 4096^2, r = 512 = 165 TFLOP/s — every chunk iteration runs at the memory latency, the loads have the
    // 16 MFMAs of one chunk to arrive.  A second register set with the loads issued two chunks ahead needs 256 VGPRs + 3.8 KB of scratch per lane
    // as written here — the per-row clamped 64-bit addresses — and was taken out again: no kernel of this library uses scratch.)
    uint32_t ra[16], ra2[16], rb[2][2][4];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t grow = min(R0 + arg + 8 * j, m - 1);
            const int64_t d0 = ((grow * r + k0) >> 1) + al;
            ra[j] = A32[min(d0, last_dw)];
            if (r_odd) ra2[j] = A32[min(d0 + 1, last_dw)];
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t k = k0 + 2 * (bp4 + 16 * jj) + q;
                u32x4 x = (u32x4){0u, 0u, 0u, 0u};
                if (k < r && bcol_ok) x = *(const u32x4*)(B + k * n + bcol);
                rb[jj][q][0] = x[0]; rb[jj][q][1] = x[1]; rb[jj][q][2] = x[2]; rb[jj][q][3] = x[3];
            }
    };
    auto stash = [&](int set, int64_t k0) {
        // A: one word (k = k0 + 2 l, + 1) of 16 rows -> [row][2 l]; consecutive lanes write consecutive words
        const int64_t k = k0 + 2 * al;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = arg + 8 * j;
            const int64_t grow = R0 + row;
            uint32_t v = ra[j];
            if (r_odd) {
                const int par = (int)((min(grow, m - 1) * r + k0) & 1);
                v = __builtin_amdgcn_alignbyte(ra2[j], ra[j], 2 * par);
            }
            if (grow >= m || k >= r) v = 0u;
            else if (k + 1 >= r) v &= 0x0000ffffu;
            *(uint32_t*)(img[set][0] + row * R16_LD + 2 * al) = v;
        }
        // B: (k, k + 1) of 8 columns -> one word per column at [col][2 p]
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int p_ = bp4 + 16 * jj;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = __builtin_amdgcn_perm(rb[jj][1][j], rb[jj][0][j], 0x05040100u);   // {row1.lo16 : row0.lo16} = column 2j
                const uint32_t hi = __builtin_amdgcn_perm(rb[jj][1][j], rb[jj][0][j], 0x07060302u);   // {row1.hi16 : row0.hi16} = column 2j + 1
                *(uint32_t*)(img[set][1] + (8 * bcl + 2 * j) * R16_LD + 2 * p_) = lo;
                *(uint32_t*)(img[set][1] + (8 * bcl + 2 * j + 1) * R16_LD + 2 * p_) = hi;
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    const int64_t nit = (r + 63) / 64;
    auto mma = [&](int set) __attribute__((always_inline)) {
        const uint16_t* ia = img[set][0] + (64 * wr + c) * R16_LD + 8 * h;
        const uint16_t* ib = img[set][1] + (64 * wc + c) * R16_LD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *(const u32x4*)(ia + 32 * t * R16_LD + 16 * ks);
                b[t] = *(const u32x4*)(ib + 32 * t * R16_LD + 16 * ks);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a[i]), __builtin_bit_cast(b16x8, b[j]), acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a[i]), __builtin_bit_cast(h16x8, b[j]), acc[i][j], 0, 0, 0);
                }
        }
    };
    fetch(0);
    stash(0, 0);
    __syncthreads();
    for (int64_t it = 0; it < nit; ++it) {
        const int set = (int)(it & 1);
        if (it + 1 < nit) fetch((it + 1) * 64);
        mma(set);
        if (it + 1 < nit) stash(set ^ 1, (it + 1) * 64);   // the other set: its last readers passed the barrier of the previous iteration
        __syncthreads();
    }
    const int64_t r0 = R0 + wr * 64, c0 = C0 + wc * 64;
    double e2 = 0.0, w2 = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t row = r0 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h, col = c0 + 32 * j + c;
                if (row < m && col < n) {
                    const float wv = elem<WT>::ld(W, row * ldw + col);
                    const double d = (double)wv - (double)acc[i][j][reg];
                    e2 += d * d;
                    w2 += (double)wv * (double)wv;
                }
            }
    e2 = wave_reduce_sum_d(e2);
    w2 = wave_reduce_sum_d(w2);
    __shared__ double red[4][2];
    if (lane == 0) { red[w][0] = e2; red[w][1] = w2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        part[2 * bid + 0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        part[2 * bid + 1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    }
}
__global__ __launch_bounds__(256) void ordered_sum_d2_kernel(const double* __restrict__ part, int64_t np, double* __restrict__ out) {
    __shared__ double red[256][2];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < np; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    red[threadIdx.x][0] = a;
    red[threadIdx.x][1] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x][0] += red[threadIdx.x + o][0]; red[threadIdx.x][1] += red[threadIdx.x + o][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[0][1]; }
}

int absstat_nsplit(int64_t rows, int64_t cols) {
    // enough workgroups to fill 256 CUs, at least 32 rows per split
    const int64_t colblocks = ceil_div64(cols, 64 * 4);
    int64_t ns = ceil_div64(1024, colblocks);
    const int64_t maxs = rows / 32 > 0 ? rows / 32 : 1;
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return (int)ns;
}

}  // namespace

extern "C" {

int asvd_version(void) { return 100; }

const char* asvd_status_string(int status) {
    switch (status) {
        case ASVD_OK: return "ok";
        case ASVD_E_BADARG: return "bad argument";
        case ASVD_E_WORKSPACE: return "workspace too small";
        case ASVD_E_HIP: return "HIP runtime error";
        case ASVD_E_NODEVICE: return "no gfx950 device";
        case ASVD_N_NOCONV: return "Jacobi SVD did not converge within max_sweeps";
        case ASVD_N_NAN: return "NaN/Inf encountered";
        default: return "unknown status";
    }
}

int asvd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int asvd_absstat_worksize(int64_t rows, int64_t cols, size_t* bytes) {
    if (!bytes || rows < 1 || cols < 1) return ASVD_E_BADARG;
    const int ns = absstat_nsplit(rows, cols);
    *bytes = (size_t)ns * cols * (sizeof(float) + sizeof(int));
    return ASVD_OK;
}

// the two halves of asvd_absstat_accum: the pass over X (partials into `work`) and the ordered finalize into ONE accumulator.  Linears
// that receive the same input tensor (q/k/v, gate/up) share the first half: X is read once per distinct input.
int asvd_absstat_partial(const void* x, int x_dtype, int64_t rows, int64_t cols, int64_t ld, int mode, void* work, size_t work_bytes,
                         void* stream) {
    if (!x || !work || rows < 1 || cols < 1 || ld < cols || !dtype_ok(x_dtype)) return ASVD_E_BADARG;
    if (mode != ASVD_STAT_ABS_MEAN && mode != ASVD_STAT_ABS_MAX && mode != ASVD_STAT_SQ_MEAN) return ASVD_E_BADARG;
    const int ns = absstat_nsplit(rows, cols);
    if (work_bytes < (size_t)ns * cols * (sizeof(float) + sizeof(int))) return ASVD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)work;
    int* nanflag = (int*)(part + (int64_t)ns * cols);
    const int64_t rps = ceil_div64(rows, ns);
    const int vec = (x_dtype == ASVD_F32) ? 4 : 8;
    dim3 grid((unsigned)ceil_div64(cols, 64 * vec), (unsigned)ns);
    ASVD_DISPATCH_DTYPE(x_dtype, XT, {
        if (mode == ASVD_STAT_ABS_MEAN) absstat_partial_kernel<XT, ASVD_STAT_ABS_MEAN><<<grid, 256, 0, st>>>(x, rows, cols, ld, rps, part, nanflag);
        else if (mode == ASVD_STAT_SQ_MEAN) absstat_partial_kernel<XT, ASVD_STAT_SQ_MEAN><<<grid, 256, 0, st>>>(x, rows, cols, ld, rps, part, nanflag);
        else absstat_partial_kernel<XT, ASVD_STAT_ABS_MAX><<<grid, 256, 0, st>>>(x, rows, cols, ld, rps, part, nanflag);
    });
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_absstat_finalize(const void* work, size_t work_bytes, int64_t rows, int64_t cols, void* acc, int acc_dtype, int mode, void* stream) {
    if (!work || !acc || rows < 1 || cols < 1 || !dtype_ok(acc_dtype)) return ASVD_E_BADARG;
    if (mode != ASVD_STAT_ABS_MEAN && mode != ASVD_STAT_ABS_MAX && mode != ASVD_STAT_SQ_MEAN) return ASVD_E_BADARG;
    const int ns = absstat_nsplit(rows, cols);
    if (work_bytes < (size_t)ns * cols * (sizeof(float) + sizeof(int))) return ASVD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const float* part = (const float*)work;
    const int* nanflag = (const int*)(part + (int64_t)ns * cols);
    const unsigned fg = (unsigned)ceil_div64(cols, 64);
    ASVD_DISPATCH_DTYPE(acc_dtype, AT, {
        if (mode != ASVD_STAT_ABS_MAX) absstat_final_kernel<AT, ASVD_STAT_ABS_MEAN><<<fg, 256, 0, st>>>(part, nanflag, ns, rows, cols, acc);
        else absstat_final_kernel<AT, ASVD_STAT_ABS_MAX><<<fg, 256, 0, st>>>(part, nanflag, ns, rows, cols, acc);
    });
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_absstat_accum(const void* x, int x_dtype, int64_t rows, int64_t cols, int64_t ld, void* acc, int acc_dtype, int mode,
                       void* work, size_t work_bytes, void* stream) {
    if (!acc || !dtype_ok(acc_dtype)) return ASVD_E_BADARG;
    const int rc = asvd_absstat_partial(x, x_dtype, rows, cols, ld, mode, work, work_bytes, stream);
    if (rc) return rc;
    return asvd_absstat_finalize(work, work_bytes, rows, cols, acc, acc_dtype, mode, stream);
}

int asvd_make_scale(const void* scaling, const void* fisher, int dtype, int64_t n, float alpha, float eps, void* out, void* stream) {
    if (!scaling || !out || n < 1 || !dtype_ok(dtype)) return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = (unsigned)ceil_div64(n, 256);
    ASVD_DISPATCH_DTYPE(dtype, DT, { make_scale_kernel<DT><<<g, 256, 0, st>>>(scaling, fisher, n, alpha, eps, out); });
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_scale_cols(const void* w, int w_dtype, int64_t m, int64_t n, int64_t ldw, const void* s, int s_dtype, float* out,
                    int64_t ldo, void* stream) {
    if (!w || !out || m < 1 || n < 1 || ldw < n || ldo < n || !dtype_ok(w_dtype)) return ASVD_E_BADARG;
    if (s && !dtype_ok(s_dtype)) return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)ceil_div64(n, 256), (unsigned)ceil_div64(m, 32));
    const int has = s ? 1 : 0;
    const int sdt = s ? s_dtype : ASVD_F32;
    ASVD_DISPATCH_DTYPE(w_dtype, WT, {
        switch (sdt) {
            case ASVD_F32: scale_cols_kernel<WT, ASVD_F32><<<grid, 256, 0, st>>>(w, m, n, ldw, s, has, out, ldo); break;
            case ASVD_F16: scale_cols_kernel<WT, ASVD_F16><<<grid, 256, 0, st>>>(w, m, n, ldw, s, has, out, ldo); break;
            default: scale_cols_kernel<WT, ASVD_BF16><<<grid, 256, 0, st>>>(w, m, n, ldw, s, has, out, ldo); break;
        }
    });
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_truncate_split(const float* U, int64_t ldu, const float* S, const float* V, int64_t ldv, const void* s, int s_dtype,
                        int64_t m, int64_t n, int64_t r, int sigma_fuse, void* A, void* B, int out_dtype, int* nan_flags,
                        void* stream) {
    if (!U || !S || !V || !A || !B || m < 1 || n < 1 || r < 1 || ldu < r || ldv < r || !dtype_ok(out_dtype)) return ASVD_E_BADARG;
    if (sigma_fuse < 0 || sigma_fuse > 2) return ASVD_E_BADARG;
    if (s && !dtype_ok(s_dtype)) return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int has = s ? 1 : 0;
    const int sdt = s ? s_dtype : ASVD_F32;
    dim3 ga((unsigned)ceil_div64(r, 256), (unsigned)ceil_div64(m, 32));
    dim3 gb((unsigned)ceil_div64(n, 32), (unsigned)ceil_div64(r, 32));
    ASVD_DISPATCH_DTYPE(out_dtype, OT, {
        split_a_kernel<OT><<<ga, 256, 0, st>>>(U, ldu, S, m, r, sigma_fuse, A, nan_flags);
        switch (sdt) {
            case ASVD_F32: split_b_kernel<OT, ASVD_F32><<<gb, 256, 0, st>>>(V, ldv, S, s, has, n, r, sigma_fuse, B, nan_flags); break;
            case ASVD_F16: split_b_kernel<OT, ASVD_F16><<<gb, 256, 0, st>>>(V, ldv, S, s, has, n, r, sigma_fuse, B, nan_flags); break;
            default: split_b_kernel<OT, ASVD_BF16><<<gb, 256, 0, st>>>(V, ldv, S, s, has, n, r, sigma_fuse, B, nan_flags); break;
        }
    });
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_fro_worksize(int64_t m, int64_t n, size_t* bytes) {
    if (!bytes || m < 1 || n < 1) return ASVD_E_BADARG;
    *bytes = (size_t)ceil_div64(m, 8) * sizeof(float);
    return ASVD_OK;
}

int asvd_fro_norm_sq(const void* w, int w_dtype, int64_t m, int64_t n, int64_t ldw, float* out, void* work, size_t work_bytes,
                     void* stream) {
    if (!w || !out || !work || m < 1 || n < 1 || ldw < n || !dtype_ok(w_dtype)) return ASVD_E_BADARG;
    const int64_t np = ceil_div64(m, 8);
    if (work_bytes < (size_t)np * sizeof(float)) return ASVD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    ASVD_DISPATCH_DTYPE(w_dtype, WT, { sumsq_partial_kernel<WT><<<(unsigned)np, 256, 0, st>>>(w, m, n, ldw, (float*)work); });
    ordered_sum_f32_kernel<<<1, 256, 0, st>>>((const float*)work, np, out);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

static size_t recon_part_bytes(int64_t m, int64_t n) { return (((size_t)(ceil_div64(m, 64) * ceil_div64(n, 64)) * 2 * sizeof(double)) + 255) & ~(size_t)255; }

int asvd_reconstruct_worksize(int64_t m, int64_t n, int64_t r, size_t* bytes) {
    if (!bytes || m < 1 || n < 1 || r < 1) return ASVD_E_BADARG;
    const int64_t rp = round_up64(r, 64);
    *bytes = recon_part_bytes(m, n) + (size_t)(m + n) * rp * 2;   // partial sums + the K-contiguous padded copies of 16-bit factors
    return ASVD_OK;
}

int asvd_reconstruct_err(const void* W, int w_dtype, int64_t ldw, const void* A, const void* B, int ab_dtype, int64_t m, int64_t n,
                         int64_t r, double* out, void* work, size_t work_bytes, void* stream) {
    if (!W || !A || !B || !out || !work || m < 1 || n < 1 || r < 1 || ldw < n || !dtype_ok(w_dtype) || !dtype_ok(ab_dtype)) return ASVD_E_BADARG;
    size_t need = 0;
    (void)asvd_reconstruct_worksize(m, n, r, &need);
    if (work_bytes < need) return ASVD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)work;
    if (ab_dtype == ASVD_F32) {  // fp32 factors (not what SVDLinear holds): one wave per 32 x 32 tile on the fp32 MFMA
        const int64_t gx = ceil_div64(n, 64), gy = ceil_div64(m, 64);
        dim3 grid((unsigned)gx, (unsigned)gy);
        ASVD_DISPATCH_DTYPE(w_dtype, WT, { recon_err_kernel<WT, ASVD_F32><<<grid, 256, 0, st>>>(W, ldw, A, B, m, n, r, part); });
        ordered_sum_d2_kernel<<<1, 256, 0, st>>>(part, gx * gy, out);
        ASVD_HIP_CHECK(hipGetLastError());
        return ASVD_OK;
    }
    const int64_t gx16 = ceil_div64(n, 128), gy16 = ceil_div64(m, 128);
    if ((n % 8) == 0 && (((uintptr_t)B) & 15) == 0 && (((uintptr_t)A) & 3) == 0) {
        // one GEMM launch straight from the factors as stored (recon_err16_fused_kernel) + the ordered sum of its partials
        dim3 grid16((unsigned)gx16, (unsigned)gy16);
        ASVD_DISPATCH_DTYPE(w_dtype, WT, {
            if (ab_dtype == ASVD_BF16) recon_err16_fused_kernel<WT, 1><<<grid16, 256, 0, st>>>(W, ldw, (const uint16_t*)A, (const uint16_t*)B, m, n, r, part);
            else recon_err16_fused_kernel<WT, 0><<<grid16, 256, 0, st>>>(W, ldw, (const uint16_t*)A, (const uint16_t*)B, m, n, r, part);
        });
        ordered_sum_d2_kernel<<<1, 256, 0, st>>>(part, gx16 * gy16, out);
        ASVD_HIP_CHECK(hipGetLastError());
        return ASVD_OK;
    }
    // in_features not a multiple of 8 (or an unaligned view): K-contiguous padded copies first, then the GEMM that reads them from global memory
    const int64_t rp = round_up64(r, 64);
    uint16_t* Ap = (uint16_t*)((char*)work + recon_part_bytes(m, n));
    uint16_t* Bt = Ap + m * rp;
    pad_rows16_kernel<<<dim3((unsigned)m), 256, 0, st>>>((const uint16_t*)A, m, r, rp, Ap);
    transpose_pad16_kernel<<<dim3((unsigned)ceil_div64(n, 32), (unsigned)(rp / 32)), 256, 0, st>>>((const uint16_t*)B, r, n, rp, Bt);
    const int64_t gx = ceil_div64(n, 128), gy = ceil_div64(m, 128);
    dim3 grid((unsigned)gx, (unsigned)gy);
    ASVD_DISPATCH_DTYPE(w_dtype, WT, {
        if (ab_dtype == ASVD_BF16) recon_err16_kernel<WT, 1><<<grid, 256, 0, st>>>(W, ldw, Ap, Bt, m, n, rp, part);
        else recon_err16_kernel<WT, 0><<<grid, 256, 0, st>>>(W, ldw, Ap, Bt, m, n, rp, part);
    });
    ordered_sum_d2_kernel<<<1, 256, 0, st>>>(part, gx * gy, out);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_make_scale_batched(int batch, const void* const* scaling_host, const void* const* fisher_host, int dtype, int64_t n, float alpha,
                            float eps, void* const* out_host, void* stream) {
    if (batch < 1 || !scaling_host || !out_host) return ASVD_E_BADARG;
    for (int b = 0; b < batch; ++b) {
        const int rc = asvd_make_scale(scaling_host[b], fisher_host ? fisher_host[b] : nullptr, dtype, n, alpha, eps, out_host[b], stream);
        if (rc) return rc;
    }
    return ASVD_OK;
}

int asvd_truncate_split_batched(int batch, const float* const* U_host, int64_t ldu, const float* const* S_host, const float* const* V_host,
                                int64_t ldv, const void* const* s_host, int s_dtype, int64_t m, int64_t n, int64_t r, int sigma_fuse,
                                void* const* A_host, void* const* B_host, int out_dtype, int* nan_flags, void* stream) {
    if (batch < 1 || !U_host || !S_host || !V_host || !A_host || !B_host) return ASVD_E_BADARG;
    for (int b = 0; b < batch; ++b) {
        const int rc = asvd_truncate_split(U_host[b], ldu, S_host[b], V_host[b], ldv, s_host ? s_host[b] : nullptr, s_dtype, m, n, r, sigma_fuse,
                                           A_host[b], B_host[b], out_dtype, nan_flags ? nan_flags + 3 * b : nullptr, stream);
        if (rc) return rc;
    }
    return ASVD_OK;
}

}  // extern "C"
