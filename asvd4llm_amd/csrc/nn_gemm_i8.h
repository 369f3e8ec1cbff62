// nn_gemm_i8.h — the long-side product Y = X Vr of the tall path's epilogue on the int8 matrix pipe (included by svd_jacobi.hip inside its anonymous
// namespace, after gram_i8.h).  Round 6.
//
// nn_gemm_split_kernel (tall_kernels.h) forms every fp32 product from six bf16 products with fp32 accumulation and splits its operands inside the GEMM
// loop.  Here both operands become 24-bit fixed-point numbers ONCE, outside the loop, and the products are integer:
//   x[r][c] ~ tx[r][c] 2^(alpha_r + beta_c - 23),   v[c][j] ~ tv[c][j] 2^(gamma_j - beta_c - 23),   |tx|, |tv| <= 127 * 2^16 (three balanced radix-256 digits)
//   beta_c  : power of two just above the column norm of X (removes the activation scales from the row maxima; cancels in the product),
//   alpha_r : power of two just above max_c |x[r][c]| 2^-beta_c,     gamma_j : just above max_c |v[c][j]| 2^beta_c,
//   y[r][j] = 2^(alpha_r + gamma_j - 46) sum_c tx tv,   sum_c tx tv = sum_{a,b} 2^(8 (4 - a - b)) (Da^T Db)[r][j]
// with the eight digit products of weight s = a + b <= 3 kept (four int32 accumulators: exact, no accumulation rounding) and only the product of the two
// lowest digits dropped, 2^-32 of the largest product.  v_mfma_i32_32x32x32_i8 covers twice the reduction length of v_mfma_f32_32x32x16_bf16 per issue
// slot, the operands are half the bytes and need no VALU in the loop.  Reduction lengths (columns of X) above 32768 keep the bf16 kernel (int32 range).
// What is left is the rounding of the fixed-point operands themselves, 2^-25 of their ROW (X) / COLUMN (V) maximum per entry — the three bf16 parts of the
// other form represent every entry exactly, that form's error is its fp32 accumulation over the reduction; measured side by side in
// tests/test_gpu_gram_i8.py (columns of U sigma against the fp64 product).  ASVD_NN_I8=0 keeps the bf16 kernel.
#pragma once

// bex[b][c] = beta_c from the column norms d of the reduction (any power of two above the column's largest entry does; |x| <= |x_c|_2 < 2^beta)
__global__ void colexp_from_norm_kernel(const double* __restrict__ d, int n, int* __restrict__ bex) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = d[i];
    int e = GI_ZERO;
    if (!(v == v) || v > 1.7e308) e = GI_BAD;
    else if (v > 0.0) { (void)frexp(v, &e); }  // v = f 2^e, f in [0.5, 1)
    bex[i] = e;
}

// Maxima of |x| 2^shift over a row / column are taken on KEYS (exponent + shift in the high bits, mantissa below: ordered like the scaled magnitudes,
// with no intermediate overflow or underflow); 0 = all zero, 0xffffffff = Inf / NaN met.
__device__ __forceinline__ unsigned gi_key(float x, int shift) {
    const unsigned a = __float_as_uint(x) & 0x7fffffffu;
    if (a >= 0x7f800000u) return 0xffffffffu;
    if (a == 0u) return 0u;
    int e;
    const float f = frexpf(__uint_as_float(a), &e);                 // f in [0.5, 1)
    const unsigned mant = (unsigned)ldexpf(f, 24) - 0x800000u;      // 23 bits, exact
    const int ee = min(max(e + shift + 256, 1), 510);                // nine bits; beyond 2^+-254 of the scale the order no longer matters
    return ((unsigned)ee << 23) | mant;
}
__device__ __forceinline__ int gi_exp_of_key(unsigned key) {       // E with (max) 2^-E < 127/128
    if (key == 0xffffffffu) return GI_BAD;
    if (key == 0u) return GI_ZERO;
    return (int)(key >> 23) - 256 + ((key & 0x7fffffu) >= 0x7e0000u ? 1 : 0);
}

// aex[z][r] = alpha_r.  grid (ceil(rows / 8), zb), 256 threads: eight rows x 32 lanes over the columns of a panel.
__global__ __launch_bounds__(256) void rowmaxexp_kernel(const float* __restrict__ Xall, int64_t panel_stride, int64_t batch_stride, int nb, int rows, int rows_pad,
                                                        const int* __restrict__ bex, int n_pad, int* __restrict__ aex) {
    const int z = blockIdx.y, r = blockIdx.x * 8 + (threadIdx.x >> 5), c = threadIdx.x & 31;
    if (r >= rows_pad) return;
    unsigned mx = 0;
    if (r < rows) {
        const float* __restrict__ xp = Xall + (int64_t)z * batch_stride + (int64_t)r * PB + c;
        const int* __restrict__ be = bex + (int64_t)z * n_pad + c;
        for (int P = 0; P < nb; ++P) {
            const int b_ = be[P * PB];
            const unsigned a = b_ == GI_ZERO ? 0u : (b_ == GI_BAD ? 0xffffffffu : gi_key(xp[(int64_t)P * panel_stride], -b_));
            mx = mx > a ? mx : a;
        }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)mx, o, 32);
        mx = mx > t ? mx : t;
    }
    if (c == 0) aex[(int64_t)z * rows_pad + r] = gi_exp_of_key(mx);
}

// Digit planes of X^T for the rows [r0, r0 + 32 rps):  planes[z][digit][row panel RP][16-column group cg][row rr][16 bytes = columns].
// grid (rps, ceil(cgs / 8), zb), 256 threads = 8 column groups x 32 rows.
__global__ __launch_bounds__(256) void split_xt_i8_kernel(const float* __restrict__ Xall, int64_t panel_stride, int64_t batch_stride, int rows, int rows_pad,
                                                          const int* __restrict__ bex, int n_pad, const int* __restrict__ aex, int r0, int cgs,
                                                          signed char* __restrict__ planes, int64_t plane_stride) {
    const int RP = blockIdx.x, z = blockIdx.z, rr = threadIdx.x & 31, cg = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (cg >= cgs) return;
    const int r = r0 + RP * 32 + rr;
    unsigned w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
    const int ae = r < rows ? aex[(int64_t)z * rows_pad + r] : GI_ZERO;
    if (ae != GI_ZERO) {
        const float* __restrict__ xp = Xall + (int64_t)z * batch_stride + (int64_t)(cg >> 1) * panel_stride + (int64_t)r * PB + 16 * (cg & 1);
        const int* __restrict__ be = bex + (int64_t)z * n_pad + cg * 16;
        float x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *(const float4*)(xp + 4 * q);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // a row that holds Inf / NaN (ae == GI_BAD): digits of no meaning, the epilogue writes NaN for the row
            const int t = (ae == GI_BAD || be[i] == GI_ZERO || be[i] == GI_BAD) ? 0 : (int)rintf(ldexpf(x[i], 23 - be[i] - ae));
            const int d2 = (int)(signed char)(t & 0xff);
            const int t1 = (t - d2) >> 8;
            const int d1 = (int)(signed char)(t1 & 0xff);
            const int d0 = (t1 - d1) >> 8;
            w0[i >> 2] |= (unsigned)(d0 & 0xff) << (8 * (i & 3));
            w1[i >> 2] |= (unsigned)(d1 & 0xff) << (8 * (i & 3));
            w2[i >> 2] |= (unsigned)(d2 & 0xff) << (8 * (i & 3));
        }
    }
    signed char* pb = planes + (int64_t)z * 3 * plane_stride + (((int64_t)RP * cgs + cg) * 32 + rr) * 16;
    *(uint4*)(pb) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
    *(uint4*)(pb + plane_stride) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    *(uint4*)(pb + 2 * plane_stride) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
}

// vmax[z][j] = key of max_c |v[c][j]| 2^beta_c  (pre-zeroed; grid (ceil(k / 256), row chunks, zb), atomic max)
__global__ __launch_bounds__(256) void vcolmax_kernel(TallBatch tb, int cols, int k, int64_t ldv, const int* __restrict__ bex, int n_pad, int rows_per_chunk,
                                                      unsigned* __restrict__ vmax, int kp) {
    const int z = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k) return;
    const float* __restrict__ V = tb.vr[z];
    const int c0 = blockIdx.y * rows_per_chunk, c1 = min(c0 + rows_per_chunk, cols);
    unsigned mx = 0;
    for (int c = c0; c < c1; ++c) {
        const int be = bex[(int64_t)z * n_pad + c];
        const unsigned a = be == GI_ZERO ? 0u : (be == GI_BAD ? 0xffffffffu : gi_key(V[(int64_t)c * ldv + j], be));
        mx = mx > a ? mx : a;
    }
    atomicMax(vmax + (int64_t)z * kp + j, mx);
}
// Digit planes of Vr:  planes[z][digit][column panel JP][16-row group cg][column jj][16 bytes = rows c];  gex[z][j] = gamma_j.
// grid (kp / 32, ceil(cgs / 8), zb), 256 threads = 8 row groups x 32 columns.
__global__ __launch_bounds__(256) void split_v_i8_kernel(TallBatch tb, int cols, int k, int64_t ldv, const int* __restrict__ bex, int n_pad,
                                                         const unsigned* __restrict__ vmax, int kp, int cgs, signed char* __restrict__ planes,
                                                         int64_t plane_stride, int* __restrict__ gex) {
    const int JP = blockIdx.x, z = blockIdx.z, jj = threadIdx.x & 31, cg = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (cg >= cgs) return;
    const int j = JP * 32 + jj;
    const int ge = j < k ? gi_exp_of_key(vmax[(int64_t)z * kp + j]) : GI_ZERO;
    if (cg == 0) gex[(int64_t)z * kp + j] = ge;
    unsigned w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
    if (ge != GI_ZERO && ge != GI_BAD) {
        const float* __restrict__ V = tb.vr[z];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cg * 16 + i;
            int t = 0;
            if (c < cols) {
                const int be = bex[(int64_t)z * n_pad + c];
                if (be != GI_ZERO && be != GI_BAD) t = (int)rintf(ldexpf(V[(int64_t)c * ldv + j], 23 + be - ge));
            }
            const int d2 = (int)(signed char)(t & 0xff);
            const int t1 = (t - d2) >> 8;
            const int d1 = (int)(signed char)(t1 & 0xff);
            const int d0 = (t1 - d1) >> 8;
            w0[i >> 2] |= (unsigned)(d0 & 0xff) << (8 * (i & 3));
            w1[i >> 2] |= (unsigned)(d1 & 0xff) << (8 * (i & 3));
            w2[i >> 2] |= (unsigned)(d2 & 0xff) << (8 * (i & 3));
        }
    }
    signed char* pb = planes + (int64_t)z * 3 * plane_stride + (((int64_t)JP * cgs + cg) * 32 + jj) * 16;
    *(uint4*)(pb) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
    *(uint4*)(pb + plane_stride) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    *(uint4*)(pb + 2 * plane_stride) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
}

// out[z][r0 + ..][..] = Y.  Workgroup = 512 threads = one 128 (rows) x 128 (columns) block on the tiling of gram_i8_kernel: wave w owns row panels
// 4 Ib + 2 (w >> 2) + {0, 1} against column panel 4 Jb + (w & 3) — 2 tiles x 4 weights x 16 = 128 accumulators, two waves per SIMD; stages of 64 reduction
// indices through LDS in the planes' own order.  grid: (column blocks x row blocks of this segment, zb); `order` 1 walks the blocks in 4 x 8 groups per XCD,
// 0 in row runs per XCD, 2 plainly (measured: no difference, tools/bench_i8_gemm.py).
// (First version, measured: four waves of 64 x 64 with SIX products, weight <= 2 — 17 ms per 32 x 4096^2, but on rows with a spike over a floor 1e-3
// below it the floor's values sit in the low digits and the dropped weight-3 products were 2.4e-5 of sigma_1 in U sigma, fifty times the bf16 form.)
__global__ __launch_bounds__(512) void nn_gemm_i8_kernel(TallBatch tb, const signed char* __restrict__ planesA, int64_t strideA, int rps,
                                                         const signed char* __restrict__ planesB, int64_t strideB, int jps, int cgs,
                                                         const int* __restrict__ aex, int rows_pad, const int* __restrict__ gex, int kp, int r0, int rows, int k,
                                                         int64_t ldo, int gx, int gy, int order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gi_lds[];  // 2 x GI_STAGE_BYTES
    const int z = blockIdx.y;
    float* __restrict__ out = tb.lng[z];
    if (!out) return;
    int Ib, Jb;
    {
        const int total = gx * gy, id = blockIdx.x;
        if (order == 1 && (total & 7) == 0 && (gx & 7) == 0 && (gy & 3) == 0) {
            const int lid = (id & 7) * (total >> 3) + (id >> 3);   // consecutive ids of one XCD
            const int g = lid >> 5, wi_ = lid & 31, ggx = gx >> 3;
            Ib = (g / ggx) * 4 + (wi_ >> 3);
            Jb = (g % ggx) * 8 + (wi_ & 7);
        } else if (order == 0 && (total & 7) == 0) {
            const int lid = (id & 7) * (total >> 3) + (id >> 3);   // row runs per XCD
            Ib = lid / gx;
            Jb = lid % gx;
        } else {
            Ib = id / gx;
            Jb = id % gx;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w >> 2, wj = w & 3;
    const unsigned char* gsrc[6];
    bool gok[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int q = r * 512 + tid;
        const int side = q / 1536, qq = q % 1536;
        const int a = (qq >> 7) >> 2, p4 = (qq >> 7) & 3, within = qq & 127;
        if (side == 0) {
            const int P = 4 * Ib + p4;
            gok[r] = P < rps;
            gsrc[r] = (const unsigned char*)planesA + (int64_t)z * 3 * strideA + (int64_t)a * strideA + (int64_t)(gok[r] ? P : 0) * cgs * 512 + within * 16;
        } else {
            const int P = 4 * Jb + p4;
            gok[r] = P < jps;
            gsrc[r] = (const unsigned char*)planesB + (int64_t)z * 3 * strideB + (int64_t)a * strideB + (int64_t)(gok[r] ? P : 0) * cgs * 512 + within * 16;
        }
    }
    i32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][s][i] = 0;
    const int nstage = cgs >> 2;  // cgs is a multiple of 4
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
    // D[i][j]: j = lane & 31 (B operand = column panel), i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A operand = row panel)
    const int j = (4 * Jb + wj) * 32 + (lane & 31);
    if (j >= k) return;
    const int ge = gex[(int64_t)z * kp + j];
    const double sj = ge == GI_BAD ? __builtin_nan("") : (ge == GI_ZERO ? 0.0 : ldexp(1.0, ge - 23));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = r0 + (4 * Ib + 2 * wi + t) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (r >= rows) continue;
            const int ae = aex[(int64_t)z * rows_pad + r];
            const double si = ae == GI_BAD ? __builtin_nan("") : (ae == GI_ZERO ? 0.0 : ldexp(1.0, ae - 23));
            double v = (double)acc[t][3][reg] * 256.0;
            v += (double)acc[t][2][reg] * 65536.0;
            v += (double)acc[t][1][reg] * 16777216.0;
            v += (double)acc[t][0][reg] * 4294967296.0;
            out[(int64_t)r * ldo + j] = (float)(v * si * sj);
        }
    }
}
