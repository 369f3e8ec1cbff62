// sigma_max.hip — largest singular value of a batch of same-shape matrices by Lanczos on W^T W (gfx950).
//
// Replaces the values-only factorisation of calib_sensitivity_stable_rank (reference sensitivity.py:101-102:
// `_, singular_values, _ = torch.svd(w.float(), compute_uv=False); spectral_norm = torch.max(singular_values)`), which needs
// sigma_max only.  A full Jacobi SVD for one number costs ~10 passes of 8*m*n^2 flop; Lanczos needs two matrix-vector passes
// over W per step and 50-300 steps, i.e. it is bound by reading W (fp16: 2*m*n bytes per pass, L2 / Infinity-Cache resident
// for one Linear) — HBM roofline, not MFMA.
//
// Per step j (three launches, grid.y = batch):
//   lz_pass    : per 64-row chunk of W:  u = W[rows,:] v  (wave per row, fp32 accumulate), then the chunk's contribution
//                W[rows,:]^T u to w, written as a partial row [chunk][n] (no atomics: fixed summation order)
//   lz_colsum  : w[c] = sum over chunks (fixed order); per-block partials of alpha = v.w in fp64
//   lz_step    : alpha_j; w -= alpha v + beta v_prev; beta_{j+1} = |w| (fp64 accumulate); v_prev <- v; v <- w / beta
// Every 16 steps lz_ritz finds the largest eigenvalue theta of the j x j Lanczos tridiagonal by 64-way multisection on the
// Sturm count (fp64).  theta_j is monotone non-decreasing; the iteration stops when it moved by less than tol*theta over
// the last 16 steps.  No re-orthogonalisation: loss of orthogonality produces ghost copies but does not move the extreme
// Ritz value.
#include "common.h"
#include <algorithm>
#include <vector>

namespace {

constexpr int LZ_ROWS = 64;     // rows of W per workgroup in lz_pass
constexpr int LZ_CHECK = 16;    // steps between convergence checks

__device__ __forceinline__ float hash_unit(uint32_t i, uint32_t b) {  // deterministic start vector in (-1, 1), never all zero
    uint32_t x = i * 2654435761u ^ (b + 1u) * 2246822519u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    return ((float)(x & 0xffffffu) + 0.5f) * (2.0f / 16777216.0f) - 1.0f;
}

__global__ void lz_init_kernel(float* __restrict__ v, float* __restrict__ vprev, int n, int64_t vstride, double* __restrict__ beta, int max_steps) {
    const int b = blockIdx.x;
    __shared__ double red[16];
    double s = 0.0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) { float x = hash_unit((uint32_t)c, (uint32_t)b); s += (double)x * x; }
    s = wave_reduce_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    double tot = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    const float inv = (float)(1.0 / sqrt(tot));
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        v[b * vstride + c] = hash_unit((uint32_t)c, (uint32_t)b) * inv;
        vprev[b * vstride + c] = 0.0f;
    }
    if (threadIdx.x == 0) beta[(int64_t)b * (max_steps + 1)] = 0.0;
}

// 8 consecutive elements of a row as floats (16-B load for 2-byte types, 2 x 16 B for fp32); caller guarantees alignment
template <int DT> __device__ __forceinline__ void load8(const void* p, int64_t i, float (&o)[8]);
template <> __device__ __forceinline__ void load8<ASVD_F32>(const void* p, int64_t i, float (&o)[8]) {
    const f32x4 a = *(const f32x4*)((const float*)p + i), b = *(const f32x4*)((const float*)p + i + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <> __device__ __forceinline__ void load8<ASVD_F16>(const void* p, int64_t i, float (&o)[8]) {
    const f16x8 a = *(const f16x8*)((const uint16_t*)p + i);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)a[k];
}
template <> __device__ __forceinline__ void load8<ASVD_BF16>(const void* p, int64_t i, float (&o)[8]) {
    const uint4 a = *(const uint4*)((const uint16_t*)p + i);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[2 * k] = __uint_as_float(w[k] << 16); o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}

// VEC: rows are 16-byte aligned and n % 8 == 0 -> vector path; otherwise scalar loads
template <int DT, bool VEC>
__global__ __launch_bounds__(256) void lz_pass_kernel(const void* const* __restrict__ mats, int64_t lda, int m, int n,
                                                      const float* __restrict__ v, int64_t vstride,
                                                      float* __restrict__ part, int nchunks) {
    extern __shared__ float vs[];  // [n] current Lanczos vector
    __shared__ float us[LZ_ROWS];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const void* A = mats[b];
    const float* vb = v + b * vstride;
    for (int c = tid; c < n; c += 256) vs[c] = vb[c];
    __syncthreads();
    const int r0 = chunk * LZ_ROWS;
    // phase 1: u[r] = W[r,:] . v   (one wave per row, 16 rows per wave)
    for (int rr = wave; rr < LZ_ROWS; rr += 4) {
        const int r = r0 + rr;
        float acc = 0.0f;
        if (r < m) {
            if (VEC) {
                for (int c = lane * 8; c < n; c += 512) {
                    float x[8];
                    load8<DT>(A, (int64_t)r * lda + c, x);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = fmaf(x[k], vs[c + k], acc);
                }
            } else {
                for (int c = lane; c < n; c += 64) acc = fmaf(elem<DT>::ld(A, (int64_t)r * lda + c), vs[c], acc);
            }
        }
        acc = wave_reduce_sum(acc);
        if (lane == 0) us[rr] = acc;
    }
    __syncthreads();
    // phase 2: partial w[c] = sum_{r in chunk} W[r,c] u[r]   (thread per 8 columns; rows re-read from L2)
    float* pb = part + ((int64_t)b * nchunks + chunk) * vstride;
    const int rows = min(LZ_ROWS, m - r0);
    if (VEC) {
        for (int c = tid * 8; c < n; c += 2048) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int rr = 0; rr < rows; ++rr) {
                float x[8];
                load8<DT>(A, (int64_t)(r0 + rr) * lda + c, x);
                const float u = us[rr];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(x[k], u, acc[k]);
            }
            *(f32x4*)(pb + c) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *(f32x4*)(pb + c + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        }
    } else {
        for (int c = tid; c < n; c += 256) {
            float acc = 0.0f;
            for (int rr = 0; rr < rows; ++rr) acc = fmaf(elem<DT>::ld(A, (int64_t)(r0 + rr) * lda + c), us[rr], acc);
            pb[c] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void lz_colsum_kernel(const float* __restrict__ part, int nchunks, int n, int64_t vstride,
                                                        const float* __restrict__ v, float* __restrict__ w,
                                                        double* __restrict__ alpha_part, int nblk) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    __shared__ double red[4];
    double contrib = 0.0;
    if (c < n) {
        const float* p = part + (int64_t)b * nchunks * vstride + c;
        float s = 0.0f;
        for (int ch = 0; ch < nchunks; ++ch) s += p[(int64_t)ch * vstride];
        w[b * vstride + c] = s;
        contrib = (double)s * (double)v[b * vstride + c];
    }
    contrib = wave_reduce_sum_d(contrib);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) alpha_part[(int64_t)b * nblk + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void lz_step_kernel(float* __restrict__ v, float* __restrict__ vprev, const float* __restrict__ w,
                                                       int n, int64_t vstride, const double* __restrict__ alpha_part, int nblk,
                                                       double* __restrict__ alpha, double* __restrict__ beta, int max_steps, int j) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[16];
    __shared__ double sh_alpha, sh_beta;
    if (tid == 0) {
        double a = 0.0;
        for (int i = 0; i < nblk; ++i) a += alpha_part[(int64_t)b * nblk + i];
        sh_alpha = a;
        alpha[(int64_t)b * max_steps + j] = a;
    }
    __syncthreads();
    const float a = (float)sh_alpha;
    const float bj = (float)beta[(int64_t)b * (max_steps + 1) + j];
    float* vb = v + b * vstride;
    float* pb = vprev + b * vstride;
    const float* wb = w + b * vstride;
    double s = 0.0;
    // new residual kept in vprev's storage: r = w - alpha v - beta_j v_prev
    for (int c = tid; c < n; c += 1024) {
        const float r = fmaf(-bj, pb[c], fmaf(-a, vb[c], wb[c]));
        pb[c] = r;
        s += (double)r * r;
    }
    s = wave_reduce_sum_d(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        const double bn = sqrt(t);
        // invariant subspace reached (or NaN): stop growing the Krylov space; later steps see v = 0 and add zero rows to T
        const bool dead = !(bn > 1e-30 * fabs(sh_alpha) && bn > 0.0);
        sh_beta = dead ? 0.0 : bn;
        beta[(int64_t)b * (max_steps + 1) + j + 1] = (bn != bn) ? bn : sh_beta;
    }
    __syncthreads();
    const float inv = sh_beta > 0.0 ? (float)(1.0 / sh_beta) : 0.0f;
    for (int c = tid; c < n; c += 1024) {  // swap roles: vprev <- v, v <- r / beta
        const float r = pb[c], old = vb[c];
        vb[c] = r * inv;
        pb[c] = old;
    }
}

// largest eigenvalue of the j x j tridiagonal (alpha_0..alpha_{j-1}; off-diagonals beta_1..beta_{j-1}) by 64-way multisection
__global__ __launch_bounds__(64) void lz_ritz_kernel(const double* __restrict__ alpha, const double* __restrict__ beta, int max_steps, int j,
                                                     double* __restrict__ theta, float* const* __restrict__ sigma_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double* al = alpha + (int64_t)b * max_steps;
    const double* be = beta + (int64_t)b * (max_steps + 1);
    // Gershgorin bounds
    double lo = 1e300, hi = -1e300;
    bool nan = false;
    for (int i = lane; i < j; i += 64) {
        const double bl = (i > 0) ? fabs(be[i]) : 0.0, br = (i + 1 < j) ? fabs(be[i + 1]) : 0.0;
        nan |= (al[i] != al[i]) || (bl != bl) || (br != br);
        lo = fmin(lo, al[i] - bl - br);
        hi = fmax(hi, al[i] + bl + br);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
        nan |= (bool)__shfl_xor((int)nan, o, 64);
    }
    if (nan) {
        if (lane == 0) { theta[b] = __longlong_as_double(0x7ff8000000000000ll); *sigma_out[b] = __uint_as_float(0x7fc00000u); }
        return;
    }
    const double scale = fmax(fabs(lo), fabs(hi));
    const double tiny = 1e-300 + scale * 1e-30;
    for (int round = 0; round < 10 && hi - lo > 1e-15 * scale; ++round) {
        // lane l tests x_l = lo + (l+1)/65 * (hi-lo): is there an eigenvalue >= x_l, i.e. count(eigs < x_l) < j ?
        const double x = lo + (hi - lo) * (double)(lane + 1) / 65.0;
        int neg = 0;
        double q = al[0] - x;
        if (fabs(q) < tiny) q = -tiny;
        neg += q < 0.0;
        for (int i = 1; i < j; ++i) {
            q = al[i] - x - be[i] * be[i] / q;
            if (fabs(q) < tiny) q = -tiny;
            neg += q < 0.0;
        }
        const bool above = neg < j;  // lambda_max >= x
        const uint64_t mask = __ballot(above);
        // above is monotone in lane: lanes 0..p-1 true; new interval [x_{p-1}, x_p]
        const int p = __popcll(mask);
        const double nlo = (p == 0) ? lo : lo + (hi - lo) * (double)p / 65.0;
        const double nhi = (p == 64) ? hi : lo + (hi - lo) * (double)(p + 1) / 65.0;
        lo = nlo; hi = nhi;
    }
    if (lane == 0) {
        const double t = 0.5 * (lo + hi);
        theta[b] = t;
        *sigma_out[b] = (float)sqrt(fmax(t, 0.0));
    }
}

}  // namespace

extern "C" {

static int64_t lz_npad(int64_t n) { return round_up64(n, 8); }

int asvd_sigma_max_worksize(int batch, int64_t m, int64_t n, int max_steps, size_t* bytes) {
    if (!bytes || batch < 1 || m < 1 || n < 1) return ASVD_E_BADARG;
    if (max_steps <= 0) max_steps = 1024;
    max_steps = (int)round_up64(max_steps, LZ_CHECK);
    const int64_t np = lz_npad(n), nch = ceil_div64(m, LZ_ROWS), nblk = ceil_div64(n, 256);
    size_t b = 0;
    b += (size_t)batch * np * 3 * sizeof(float);              // v, vprev, w
    b += (size_t)batch * nch * np * sizeof(float);            // chunk partials
    b += (size_t)batch * nblk * sizeof(double);               // alpha partials
    b += (size_t)batch * (2 * (size_t)max_steps + 1) * sizeof(double);  // alpha, beta
    b += (size_t)batch * (sizeof(double) + 2 * sizeof(void*));  // theta, pointer tables
    *bytes = b + 1024;
    return ASVD_OK;
}

/* info_host: optional int[2*batch] = {status, lanczos steps} */
int asvd_sigma_max_batched(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                           float* const* sigma_host, int max_steps, float tol, void* work, size_t work_bytes, int* info_host,
                           void* stream) {
    if (batch < 1 || !a_host || !sigma_host || !work || m < 1 || n < 1 || lda < n || !dtype_ok(a_dtype)) return ASVD_E_BADARG;
    if (n > 16384) return ASVD_E_BADARG;  // the Lanczos vector lives in LDS (64 KB); callers use the values-only SVD beyond that
    for (int b = 0; b < batch; ++b)
        if (!a_host[b] || !sigma_host[b]) return ASVD_E_BADARG;
    if (max_steps <= 0) max_steps = 1024;
    max_steps = (int)round_up64(max_steps, LZ_CHECK);
    if (!(tol > 0.0f)) tol = 1e-8f;
    size_t need = 0;
    asvd_sigma_max_worksize(batch, m, n, max_steps, &need);
    if (work_bytes < need) return ASVD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t np = lz_npad(n);
    const int nch = (int)ceil_div64(m, LZ_ROWS), nblk = (int)ceil_div64(n, 256);
    char* p = (char*)work;
    auto take = [&](size_t bytes) { char* q = p; p += (bytes + 15) & ~(size_t)15; return (void*)q; };
    double* alpha = (double*)take((size_t)batch * max_steps * sizeof(double));
    double* beta = (double*)take((size_t)batch * (max_steps + 1) * sizeof(double));
    double* alpha_part = (double*)take((size_t)batch * nblk * sizeof(double));
    double* theta = (double*)take((size_t)batch * sizeof(double));
    const void** mats_dev = (const void**)take((size_t)batch * sizeof(void*));
    float** sig_dev = (float**)take((size_t)batch * sizeof(void*));
    float* v = (float*)take((size_t)batch * np * sizeof(float));
    float* vprev = (float*)take((size_t)batch * np * sizeof(float));
    float* w = (float*)take((size_t)batch * np * sizeof(float));
    float* part = (float*)take((size_t)batch * nch * np * sizeof(float));
    ASVD_HIP_CHECK(hipMemcpyAsync(mats_dev, a_host, (size_t)batch * sizeof(void*), hipMemcpyHostToDevice, st));
    ASVD_HIP_CHECK(hipMemcpyAsync(sig_dev, sigma_host, (size_t)batch * sizeof(void*), hipMemcpyHostToDevice, st));
    lz_init_kernel<<<batch, 256, 0, st>>>(v, vprev, (int)n, np, beta, max_steps);

    // vector path needs 16-byte aligned rows
    const size_t esz = dtype_size(a_dtype);
    bool vec = (n % 8 == 0) && ((lda * esz) % 16 == 0);
    for (int b = 0; b < batch && vec; ++b) vec = ((uintptr_t)a_host[b] % 16) == 0;
    const size_t lds = (size_t)n * sizeof(float);
    dim3 gpass((unsigned)nch, (unsigned)batch), gcol((unsigned)nblk, (unsigned)batch);

    std::vector<double> th(batch), prev(batch, -1.0);
    std::vector<int> done_at(batch, 0);
    int steps = 0, ndone = 0;
    while (steps < max_steps && ndone < batch) {
        for (int i = 0; i < LZ_CHECK; ++i, ++steps) {
            ASVD_DISPATCH_DTYPE(a_dtype, DT, {
                if (vec) lz_pass_kernel<DT, true><<<gpass, 256, lds, st>>>(mats_dev, lda, (int)m, (int)n, v, np, part, nch);
                else lz_pass_kernel<DT, false><<<gpass, 256, lds, st>>>(mats_dev, lda, (int)m, (int)n, v, np, part, nch);
            });
            lz_colsum_kernel<<<gcol, 256, 0, st>>>(part, nch, (int)n, np, v, w, alpha_part, nblk);
            lz_step_kernel<<<batch, 1024, 0, st>>>(v, vprev, w, (int)n, np, alpha_part, nblk, alpha, beta, max_steps, steps);
        }
        lz_ritz_kernel<<<batch, 64, 0, st>>>(alpha, beta, max_steps, steps, theta, sig_dev);
        ASVD_HIP_CHECK(hipGetLastError());
        ASVD_HIP_CHECK(hipMemcpyAsync(th.data(), theta, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost, st));
        ASVD_HIP_CHECK(hipStreamSynchronize(st));
        ndone = 0;
        for (int b = 0; b < batch; ++b) {
            const double t = th[b];
            const bool isnan = (t != t);
            const bool conv = isnan || (prev[b] >= 0.0 && fabs(t - prev[b]) <= (double)tol * fabs(t)) || steps >= (int)std::min<int64_t>(m, n) + LZ_CHECK;
            if (conv && !done_at[b]) done_at[b] = steps;
            if (!conv) done_at[b] = 0;
            prev[b] = t;
            ndone += done_at[b] != 0;
        }
    }
    int worst = ASVD_OK;
    for (int b = 0; b < batch; ++b) {
        int status = ASVD_OK;
        if (th[b] != th[b]) status = ASVD_N_NAN;
        else if (!done_at[b]) status = ASVD_N_NOCONV;
        if (info_host) { info_host[2 * b] = status; info_host[2 * b + 1] = done_at[b] ? done_at[b] : steps; }
        worst = std::max(worst, status);
    }
    return worst;
}

}  // extern "C"
