// snapshot_i8.h — the coupling snapshot of the sparse sweeps (fullcheck_kernel, jacobi_kernels.h) on the int8 matrix pipe (included by
// svd_jacobi.hip inside its anonymous namespace, after gram_i8.h).  Round 6.
//
// The snapshot is X^T X of the working matrix, read only through |g_ij| / sqrt(g_ii g_jj) against tol = 1e-6 (which pairs still rotate) and through
// its maximum (the termination measure).  fullcheck_kernel forms it from three fp16 products per fp32 product, splitting the operands inside the
// GEMM loop: 12.4 ms per 32 x 4096^2, bound by the L2 -> LDS operand traffic.  Here the columns are digitised once (split_i8_kernel of gram_i8.h:
// 24-bit fixed point under a power of two just above the column's LARGEST entry, measured in the pass that sums the squares) and the eight digit
// products of weight <= 3 are accumulated exactly in four int32 accumulators on the tiling of gram_i8_kernel (eight waves, 64 x 32 per wave); only
// the product of the two lowest digits is dropped, 2^-32 of the largest product.  A cosine comes out to <= ~1e-7 of the worst case (entries
// rounded to 2^-25 of the column maximum), against the 1e-6 it is compared with.
// (Measured first and WRONG: six products, weight <= 2, with exponents from the column norms — entries of a dense 4096-row column sit seven bits
// below their norm, their value lives in the two LOW digits, and the dropped weight-3 products were the second largest terms: a noise floor of
// ~5e-5 on the cosines marked nearly every pair, sparse sweeps of 129 / 107 / 92 ms instead of 37 / 24 / 17.)
// Problems of more than 32768 rows keep fullcheck_kernel (int32 range).
#pragma once

// panel_sumsq_kernel (jacobi_kernels.h) + the exponent of the column maximum in the same pass: dn[b][col] = sum of squares (fp32, same order as
// panel_sumsq_kernel), ex[b][col] = E with max |x| 2^-E < 127/128.  grid (nb, batch), 256 threads.
__global__ __launch_bounds__(256) void panel_sumsq_max_kernel(const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int m_pad, int n_pad,
                                                              float* __restrict__ dn, int* __restrict__ ex, const int* __restrict__ done) {
    const int I = blockIdx.x, b = blockIdx.y, c = threadIdx.x & 31, g = threadIdx.x >> 5;
    if (ld_flag(done + b)) return;
    const float* __restrict__ P = X + (int64_t)b * batch_stride + (int64_t)I * panel_stride;
    float s = 0.0f;
    unsigned mx = 0;
    for (int r = g; r < m_pad; r += 8) {
        const float x = P[(int64_t)r * PB + c];
        s = fmaf(x, x, s);
        const unsigned a = __float_as_uint(x) & 0x7fffffffu;
        mx = mx > a ? mx : a;
    }
    __shared__ float red[8][32];
    __shared__ unsigned rmx[8][32];
    red[g][c] = s;
    rmx[g][c] = mx;
    __syncthreads();
    if (g == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { t += red[i][c]; mx = mx > rmx[i][c] ? mx : rmx[i][c]; }
        dn[(int64_t)b * n_pad + I * PB + c] = t;
        int E;
        if (mx >= 0x7f800000u) E = GI_BAD;
        else if (mx == 0u) E = GI_ZERO;
        else {
            int e2;
            const float f = frexpf(__uint_as_float(mx), &e2);
            E = e2 + (f >= 127.0f / 128.0f ? 1 : 0);
        }
        ex[(int64_t)b * n_pad + I * PB + c] = E;
    }
}

// grid (upper-triangular 128 x 128 blocks, batch), 512 threads; marks and maximum exactly as fullcheck_kernel's epilogue.
__global__ __launch_bounds__(512) void fullcheck_i8_kernel(const signed char* __restrict__ planes, int64_t plane_stride, int nb, int kgs, int n_pad,
                                                           const int* __restrict__ ex, const float* __restrict__ dn, float tol, int kb,
                                                           unsigned char* __restrict__ pflag, unsigned* __restrict__ maxoff_bits,
                                                           const int* __restrict__ done, int nt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gi_lds[];  // 2 x GI_STAGE_BYTES
    const int b = blockIdx.y;
    if (ld_flag(done + b)) return;
    const int total = gridDim.x;
    const int lid = (total & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3));
    int Ib = 0, rem = lid;
    while (rem >= nt - Ib) { rem -= nt - Ib; ++Ib; }
    const int Jb = Ib + rem;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w >> 2, wj = w & 3;
    const signed char* __restrict__ pl = planes + (int64_t)b * 3 * plane_stride;
    const unsigned char* gsrc[6];
    bool gok[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int q = r * 512 + tid;
        const int side = q / 1536, qq = q % 1536;
        const int a = (qq >> 7) >> 2, p4 = (qq >> 7) & 3, within = qq & 127;
        const int P = 4 * (side ? Jb : Ib) + p4;
        gok[r] = P < nb;
        gsrc[r] = (const unsigned char*)pl + (int64_t)a * plane_stride + (int64_t)(gok[r] ? P : 0) * kgs * 512 + within * 16;
    }
    i32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][s][i] = 0;
    const int nstage = kgs >> 2;
    uint4 stg[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) stg[r] = gok[r] ? *(const uint4*)(gsrc[r]) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 6; ++r) *(uint4*)(gi_lds + (r * 512 + tid) * 16) = stg[r];
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const unsigned char* cur = gi_lds + (st & 1) * GI_STAGE_BYTES;
        if (st + 1 < nstage) {
#pragma unroll
            for (int r = 0; r < 6; ++r) stg[r] = gok[r] ? *(const uint4*)(gsrc[r] + (int64_t)(st + 1) * 2048) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4 fa[2][3], fb[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    fa[t][a] = *(const i32x4*)(cur + (((0 * 3 + a) * 4 + 2 * wi + t) * 4 + 2 * ks) * 512 + lane * 16);
                fb[a] = *(const i32x4*)(cur + (((1 * 3 + a) * 4 + wj) * 4 + 2 * ks) * 512 + lane * 16);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if (a + c < 4) acc[t][a + c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[t][a], fb[c], acc[t][a + c], 0, 0, 0);
        }
        if (st + 1 < nstage) {
            unsigned char* nxt = gi_lds + ((st + 1) & 1) * GI_STAGE_BYTES;
#pragma unroll
            for (int r = 0; r < 6; ++r) *(uint4*)(nxt + (r * 512 + tid) * 16) = stg[r];
        }
        __syncthreads();
    }
    const int J = 4 * Jb + wj;
    if (J >= nb) return;
    const int h = lane >> 5, c = lane & 31;
    const float* __restrict__ dnb = dn + (int64_t)b * n_pad;
    const int* __restrict__ exb = ex + (int64_t)b * n_pad;
    const float dj = dnb[J * PB + c];
    const int Ej = exb[J * PB + c];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int I = 4 * Ib + 2 * wi + t;
        if (I > J || I >= nb) continue;
        float v = 0.0f, vt = 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (I == J && i == c) continue;
            const float di = dnb[I * PB + i];
            const int Ei = exb[I * PB + i];
            // sum of the kept digit products, in units of 2^8: |.| < 2^55, exact in fp64 up to its last bit; the power-of-two scales bring it back to the data
            const double sum = (double)acc[t][3][reg] + (double)acc[t][2][reg] * 256.0 + (double)acc[t][1][reg] * 65536.0 + (double)acc[t][0][reg] * 16777216.0;
            float g = 0.0f;
            if (Ei == GI_BAD || Ej == GI_BAD) g = __builtin_nanf("");
            else if (Ei != GI_ZERO && Ej != GI_ZERO) g = (float)ldexp(sum, Ei + Ej - 38);
            const float dd = di * dj;
            float x = (dd > 0.0f) ? fabsf(g) * rsqrtf(dd) : 0.0f;
            if (g != g || dd != dd) x = __builtin_nanf("");
            const float mx = fmaxf(di, dj);
            float xt = (mx > 0.0f) ? fabsf(g) / mx : 0.0f;
            if (x != x) xt = x;
            v = nanmax(v, x);
            vt = nanmax(vt, xt);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v = nanmax(v, __shfl_xor(v, o, 64));
            vt = nanmax(vt, __shfl_xor(vt, o, 64));
        }
        if (lane == 0) {
            if (v != v) {
                atomicMax(&maxoff_bits[b], 0x7fc00000u);
            } else {
                if (v >= tol) {  // a coupling inside a panel is repaired by any visit of that panel: mark its neighbour pair
                    const int A = (I == J) ? min(I, I ^ 1) : I, Bp = (I == J) ? max(I, I ^ 1) : J;
                    pflag[((int64_t)b * nb + A) * nb + Bp] = 1;
                }
                if (I < kb || J < kb) atomicMax(&maxoff_bits[b], __float_as_uint(vt));
            }
        }
    }
}
