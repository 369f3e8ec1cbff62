// tall_kernels.h — device kernels of the Cholesky-QR reduction and of the long-side epilogue (included by svd_jacobi.hip inside its anonymous
// namespace): fp64 Gram matrix, norm sort / permute / scale, blocked fp64 Cholesky (wave-local diagonal block, MFMA block row, grouped trailing
// updates), R^T to fp32, U = X V by a split-bf16 GEMM, sigma refinement.
#pragma once

// ==================================================================================================
// Tall problems (rows >= 1.5 cols): reduce to a square one first.   X = Q R  (Cholesky-QR with the Gram matrix and the
// factorisation in FP64),  R = U_R S V^T by the block Jacobi above (cols x cols instead of rows x cols per step),  left
// vectors  U = X V S^-1  by one fp32 MFMA GEMM.  FP64 keeps the squared condition number harmless: products of fp32 entries
// are exact in fp64, so R carries a relative error ~1e-16 cond(X)^2 — below fp32 eps up to cond 3e4 — and the Gram matrix is
// scaled to unit diagonal before the factorisation, which removes column scaling (the activation scales s!) from cond.
// A non-positive pivot (rank deficiency / cond too large) makes the caller fall back to the direct path.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// G[b][I*32.., J*32..] (upper blocks, I <= J) = X_I^T X_J in fp64 with v_mfma_f64_16x16x4_f64.  One wave per 32x32 block,
// whole K range (no split, no reduction: deterministic).  The 4 waves of a workgroup share panel I through L1.
// 49.5 TFLOP/s, which is what this instruction delivers here: a variant on 64x64 blocks per wave with 16-byte loads (one eighth of the
// vector-memory instructions per MFMA) ran the same 44.4 ms per 32 x 4096^2, i.e. ~100 cycles per v_mfma_f64_16x16x4 and SIMD rather
// than the 64 the 78.6 TFLOP/s figure implies (profiles/r3_fp64_mfma_rate.txt).
__global__ __launch_bounds__(256) void gram64_kernel(const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int nb,
                                                     int m_pad, double* __restrict__ G, int64_t ldg, int64_t g_batch_stride) {
    const int I = blockIdx.x, jg = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int J = jg * 4 + w;
    if (J < I || J >= nb) return;
    const int kk = lane >> 4, cc = lane & 15;
    const float* __restrict__ pi = X + (int64_t)b * batch_stride + (int64_t)I * panel_stride + kk * PB + cc;
    const float* __restrict__ pj = X + (int64_t)b * batch_stride + (int64_t)J * panel_stride + kk * PB + cc;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int r0 = 0; r0 < m_pad; r0 += 16) {  // m_pad is a multiple of 32
        float ai[4][2], bj[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t off = (int64_t)(r0 + 4 * u) * PB;
            ai[u][0] = pi[off]; ai[u][1] = pi[off + 16];
            bj[u][0] = pj[off]; bj[u][1] = pj[off + 16];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ai[u][a], (double)bj[u][c], acc[a][c], 0, 0, 0);
    }
    double* __restrict__ out = G + (int64_t)b * g_batch_stride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = I * PB + a * 16 + kk + 4 * q, col = J * PB + c * 16 + cc;
                out[(int64_t)row * ldg + col] = acc[a][c][q];
            }
}

__global__ void chol_diag_kernel(const double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int n_pad, double* __restrict__ d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (j >= n_pad) return;
    const double g = G[(int64_t)b * g_batch_stride + (int64_t)j * ldg + j];
    d[(int64_t)b * n_pad + j] = g > 0.0 ? sqrt(g) : 0.0;
}

__global__ void d_to_float_kernel(const double* __restrict__ d, int n, float* __restrict__ df) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) df[i] = (float)d[i];
}
// Gs[i][j] (i <= j) = G[perm i][perm j] / (d_perm_i d_perm_j), unit diagonal; dp[i] = d[perm[i]]
__global__ void g_permute_scale_kernel(const double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, const double* __restrict__ d,
                                       const int* __restrict__ perm, int n_pad, double* __restrict__ Gs, double* __restrict__ dp) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= n_pad || j < i) return;
    const int pi = perm[(int64_t)b * n_pad + i], pj = perm[(int64_t)b * n_pad + j];
    const double di = d[(int64_t)b * n_pad + pi], dj = d[(int64_t)b * n_pad + pj];
    if (i == 0) dp[(int64_t)b * n_pad + j] = dj;
    const int lo = pi < pj ? pi : pj, hi = pi < pj ? pj : pi;
    double v;
    if (i == j) v = 1.0;
    else v = (di > 0.0 && dj > 0.0) ? G[(int64_t)b * g_batch_stride + (int64_t)lo * ldg + hi] / (di * dj) : 0.0;
    Gs[(int64_t)b * g_batch_stride + (int64_t)i * ldg + j] = v;
}
// The epilogue kernels of the tall path take up to 32 problems per launch (blockIdx.z): the caller's outputs are separate allocations, so the
// pointers travel by value.  (One launch per problem — 128 workgroups of the long-side product on 256 CUs, 160 launches per batch of 32 — was
// 33 ms of a 571 ms bench step.)
constexpr int TALL_ZB = 32;
struct TallBatch {
    const float* vperm[TALL_ZB];   // right vectors in sorted-column order
    float* vr[TALL_ZB];            // ... in the original order
    float* lng[TALL_ZB];           // long-side vectors (nullptr: not wanted)
    float* S[TALL_ZB];
};
// out[perm[i]][:] = in[i][:]   (rows of the right vectors back to the original column order)
__global__ void row_unpermute_kernel(TallBatch tb, const int* __restrict__ perm, int64_t perm_stride, int rows, int k) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, z = blockIdx.z;
    if (c >= k || i >= rows) return;
    tb.vr[z][(int64_t)perm[z * perm_stride + i] * k + c] = tb.vperm[z][(int64_t)i * k + c];
}

constexpr int CB = 64;        // Cholesky block size
constexpr int CLD = CB + 1;   // LDS leading dimension (doubles)

// Block step jb of the right-looking upper Cholesky  G = R^T R  (in place, fp64) as two launches: the diagonal factorisation WAVE-LOCAL, one
// wave per problem, and the block row by fp64 MFMA.  (Rounds 1-2 factored the diagonal block in LDS inside every workgroup of the block row:
// ~150 of 183 us in 64 column steps of three workgroup barriers each; the Cholesky of a 4096-column Gram matrix is a chain of 64 such
// launches.)  chol_diag_wave_kernel holds the block with lane = column, registers = rows (64 doubles): a column step is two v_readlane for the pivot, two per row multiplier and one
// v_fma_f64 per remaining row, no barrier; the inverse is a back substitution per lane against R read as LDS broadcasts.  R_jj goes to the
// side buffer Dg (r_to_f32 reads the diagonal blocks there), R_jj^-1 over the block itself, where chol_trsm_kernel — one workgroup per
// block of the block row, Y = R_jj^-T G_jq with v_mfma_f64_16x16x4, wave w: rows 16 w .. 16 w + 15 — reads it.
// (Factorising inside every workgroup of the block row was measured at 384 us per launch with this wave-local form: 2048 one-wave-busy
// workgroups of 256 VGPRs and 67 KB LDS run in four rounds.)
__device__ __forceinline__ double rdlane_f64(double v, int l) {
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__global__ __launch_bounds__(64) void chol_diag_wave_kernel(double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int jb,
                                                            int* __restrict__ fail, double* __restrict__ Dg, int nbk) {
    __shared__ double Rs[CB * CLD];
    __shared__ double dinv[CB];
    const int b = blockIdx.x, lane = threadIdx.x;
    double* Gb = G + (int64_t)b * g_batch_stride;
    const int64_t o = (int64_t)jb * CB;
    double a[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) a[i] = (lane >= i) ? Gb[(o + i) * ldg + o + lane] : 0.0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        double piv = rdlane_f64(a[c], c);
        if (!(piv > 1e-13)) { bad = true; piv = 1e-13; }  // unit-diagonal scaling: pivots live in (0, 1]
        const double r = sqrt(piv), rinv = 1.0 / r;
        a[c] = (lane == c) ? r : ((lane > c) ? a[c] * rinv : 0.0);
        if (lane == c) dinv[c] = rinv;
#pragma unroll
        for (int i = c + 1; i < CB; ++i) a[i] = fma(-rdlane_f64(a[c], i), a[c], a[i]);  // lanes < i carry junk below the diagonal: never read
    }
    if (bad && lane == 0) atomicMax(&fail[b], jb + 1);
    double* dgo = Dg + ((int64_t)b * nbk + jb) * (CB * CB);
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const double v = (lane >= i) ? a[i] : 0.0;
        Rs[i * CLD + lane] = v;
        dgo[i * CB + lane] = v;
    }
    // inverse: lane j solves R z = e_j from the bottom up; z_i = 0 for i > j falls out of the masks (a wave's LDS operations complete in order)
    double z[CB];
#pragma unroll
    for (int i = CB - 1; i >= 0; --i) {
        double acc = 0.0;
#pragma unroll
        for (int k = i + 1; k < CB; ++k) acc = fma(Rs[i * CLD + k], z[k], acc);
        const double di = dinv[i];
        z[i] = (lane == i) ? di : ((lane > i) ? -acc * di : 0.0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) Gb[(o + i) * ldg + o + lane] = z[i];
}
// grid (nbk - jb - 1, batch): block q + 1 of block row jb.  Y[i][c] = sum_{k <= i} Ri[k][i] B[k][c], in place.
__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int jb) {
    const int q = blockIdx.x + 1, b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* Gb = G + (int64_t)b * g_batch_stride;
    const int64_t o = (int64_t)jb * CB, oc = (int64_t)(jb + q) * CB;
    const int kk = lane >> 4, cc = lane & 15;
    const double* __restrict__ Rip = Gb + (o + kk) * ldg + o + 16 * w + cc;
    double* Bp = Gb + (o + kk) * ldg + oc + cc;
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < 16 * (w + 1); k0 += 4) {  // wave w owns rows 16 w .. 16 w + 15: k runs to its last row only
        const double av = Rip[(int64_t)k0 * ldg];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bp[(int64_t)k0 * ldg + t * 16], acc[t], 0, 0, 0);
    }
    __syncthreads();  // every wave has read the rows of B it needs (all rows <= its own last one) before any row is overwritten
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) Gb[(o + 16 * w + kk + 4 * u) * ldg + oc + t * 16 + cc] = acc[t][u];
}

// trailing update  G_{ib,kb} -= R_{jb,ib}^T R_{jb,kb}  (jb < ib <= kb), fp64 MFMA, one workgroup per 64x64 block
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int jb, int nbk) {
    const int ib = jb + 1 + blockIdx.x, kb = jb + 1 + blockIdx.y, b = blockIdx.z;  // gridDim.x may stop short of the last block row (strip of a group)
    if (kb < ib || kb >= nbk) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double* Gb = G + (int64_t)b * g_batch_stride;
    const double* __restrict__ Ra = Gb + ((int64_t)jb * CB + kk) * ldg + (int64_t)ib * CB + w * 16 + cc;  // wave w: rows tile w of the block
    const double* __restrict__ Rb = Gb + ((int64_t)jb * CB + kk) * ldg + (int64_t)kb * CB + cc;
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k0 = 0; k0 < CB; k0 += 4) {
        const double a = Ra[(int64_t)k0 * ldg];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Rb[(int64_t)k0 * ldg + t * 16], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = (int64_t)ib * CB + w * 16 + kk + 4 * q, col = (int64_t)kb * CB + t * 16 + cc;
            Gb[row * ldg + col] -= acc[t][q];
        }
}

// The same update for a GROUP of nj finished block rows j0 .. j0+nj-1 at once (K = 64 nj): blocks (ib, kb), j0+nj <= ib <= kb.  With one
// block row per pass (chol_syrk_kernel over the whole trailing matrix) the factorisation streams the trailing matrix nbk times —
// 2.8 GB read + written per 4096-column problem, the pass was bound by that, not by the fp64 pipe (22 TFLOP/s); grouping four block
// rows makes it a quarter.  Inside a group the rows still see each other through chol_syrk_kernel restricted to the group's strip.
__global__ __launch_bounds__(256) void chol_syrk_multi_kernel(double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int j0, int nj, int nbk) {
    const int ib = j0 + nj + blockIdx.x, kb = j0 + nj + blockIdx.y, b = blockIdx.z;
    if (kb < ib || kb >= nbk) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double* Gb = G + (int64_t)b * g_batch_stride;
    const double* __restrict__ Ra = Gb + ((int64_t)j0 * CB + kk) * ldg + (int64_t)ib * CB + w * 16 + cc;
    const double* __restrict__ Rb = Gb + ((int64_t)j0 * CB + kk) * ldg + (int64_t)kb * CB + cc;
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < nj; ++j, Ra += (int64_t)CB * ldg, Rb += (int64_t)CB * ldg) {
#pragma unroll 4
        for (int k0 = 0; k0 < CB; k0 += 4) {
            const double a = Ra[(int64_t)k0 * ldg];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Rb[(int64_t)k0 * ldg + t * 16], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = (int64_t)ib * CB + w * 16 + kk + 4 * q, col = (int64_t)kb * CB + t * 16 + cc;
            Gb[row * ldg + col] -= acc[t][q];
        }
}


// The Cholesky factor as the fp32 input of the sweeps, TRANSPOSED (Jacobi runs on R^T), through 32x32 LDS tiles: both the fp64 reads and the
// fp32 stores run along rows (a column-at-a-time store: 3.4 ms per 32 x 4096^2, this: 0.9).  grid (n_pad/32, n_pad/32, batch), 256 threads.
__global__ __launch_bounds__(256) void r_to_f32_t_kernel(const double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, const double* __restrict__ Dg,
                                                         const double* __restrict__ d, int n_pad, float* __restrict__ R, int64_t r_batch_stride) {
    __shared__ float tile[32][33];
    const int tj = blockIdx.x, ti = blockIdx.y, b = blockIdx.z;  // tile rows ti (of R), columns tj
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float* Rb = R + (int64_t)b * r_batch_stride;
    if (tj < ti) {  // strictly below the block diagonal of R: zeros (its transpose is the tile (tj, ti) of R^T)
#pragma unroll
        for (int u = 0; u < 4; ++u) Rb[(int64_t)(tj * 32 + ty + 8 * u) * n_pad + ti * 32 + tx] = 0.0f;
        return;
    }
    const int j = tj * 32 + tx;
    const double dj = d[(int64_t)b * n_pad + j];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = ti * 32 + ty + 8 * u;
        double rv = 0.0;
        if (j >= i) {
            if ((i / CB) == (j / CB)) rv = Dg[((int64_t)b * (n_pad / CB) + i / CB) * (CB * CB) + (i % CB) * CB + (j % CB)];  // diagonal blocks
            else rv = G[(int64_t)b * g_batch_stride + (int64_t)i * ldg + j];
        }
        tile[ty + 8 * u][tx] = (float)(rv * dj);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) Rb[(int64_t)(tj * 32 + ty + 8 * u) * n_pad + ti * 32 + tx] = tile[tx][ty + 8 * u];
}

// out[rows, k] = X[rows, cols] * Vr[cols, k] * diag(1 / S)   — left vectors of the tall problem from the packed panels.
// Workgroup tile 128 x 128; wave w owns rows 32w..32w+31.  Split-bf16 arithmetic (twolevel.h: every fp32 operand = three bf16 exactly, six
// products per fp32 product on the bf16 matrix pipe, fp32 accumulation; the fp32-MFMA form of round 1 ran 22 ms per 16 problems).  The
// fetching thread splits its 8 consecutive k-values once and stores them as ready MFMA operands ([block][k-step][part][lane] images,
// 36-operand half blocks so the scattered A writes fall on distinct banks); a wave then issues 48 bf16 MFMAs per 32-column panel
// against 30 ds_read_b128, no VALU in the inner loop.  Same tiling (128 x 128 per workgroup, wave w = rows 32 w ..), same epilogue.
constexpr int NG_HB = 36, NG_BLK = 2 * NG_HB;
__global__ __launch_bounds__(256, 2) void nn_gemm_split_kernel(TallBatch tb, const float* __restrict__ Xall, int64_t panel_stride, int64_t batch_stride, int nb,
                                                               int rows, int cols, int64_t ldv, int k, int64_t ldo) {
    __shared__ u32x4 Aimg[4 * 2 * 3 * NG_BLK];
    __shared__ u32x4 Bimg[4 * 2 * 3 * NG_BLK];
    float* __restrict__ out = tb.lng[blockIdx.z];
    if (!out) return;
    const float* __restrict__ X = Xall + (int64_t)blockIdx.z * batch_stride;
    const float* __restrict__ Vr = tb.vr[blockIdx.z];
    const float* S = nullptr;   // unscaled product: sigma and the unit vectors follow from its column norms (colsumsq / colfinish / colscale)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
    const int h = lane >> 5, c = lane & 31;
    const int r0 = blockIdx.y * 128;
    const int c0 = blockIdx.x * 128;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x16){0};
    // A pieces: q = tid + 256 j -> row q >> 2 of the 128, k-chunk q & 3 (8 values = 32 B); B pieces: column tid & 127, k-chunk (tid >> 7) + 2 j.
    // TWO panels are in flight (register sets 0 / 1): with one, the loads of panel p + 1 had only the matrix segment of panel p (~0.8 us) to
    // arrive.  Every load is unconditional — rows, columns and the panel index are clamped instead of guarded — so that the wait counts in front
    // of a set's first use are exact (a guarded load makes the compiler wait for ALL loads): rows beyond `rows` and columns beyond `k` only reach
    // outputs that are never stored, and the K range beyond `cols` multiplies the zero padding of the packed panels.  (finalize 35.0 -> 31.0 ms per
    // 32 x 4096^2.  Measured and dropped: ONE workgroup per CU with double-buffered images and the split of panel p + 1 issued between the four
    // groups of 12 matrix instructions of panel p — 37.6 ms: a lone wave per SIMD exposes the LDS latency in front of every group.)
    f32x4 pa[2][2][2];
    float pb[2][2][8];
    int adst[2], bdst[2];
    const float* asrc[2];
    const float* bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + 256 * j, row = q >> 2, kc = q & 3;
        adst[j] = (((row >> 5) * 2 + (kc >> 1)) * 3) * NG_BLK + (kc & 1) * NG_HB + (row & 31);
        const int cc = tid & 127, kb = (tid >> 7) + 2 * j;
        bdst[j] = (((cc >> 5) * 2 + (kb >> 1)) * 3) * NG_BLK + (kb & 1) * NG_HB + (cc & 31);
        asrc[j] = X + (int64_t)min(r0 + row, rows - 1) * PB + 8 * kc;
        bsrc[j] = Vr + min(c0 + cc, k - 1);
    }
    auto fetch = [&](int p_, f32x4 (&xa)[2][2], float (&xb)[2][8]) {
        const int p = min(p_, nb - 1);
        const float* P = (const float*)0 + (int64_t)p * panel_stride;   // offset only
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* src = asrc[j] + (P - (const float*)0);
            xa[j][0] = *(const f32x4*)src;
            xa[j][1] = *(const f32x4*)(src + 4);
            const int kb = (tid >> 7) + 2 * j;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int vr = min(p * PB + 8 * kb + e, cols - 1);
                xb[j][e] = bsrc[j][(int64_t)vr * ldv];
            }
        }
    };
    auto put = [&](u32x4* dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
        u32x4 p1, p2, p3;
        unsigned x, y, z;
        split3(v0, v1, x, y, z); p1[0] = x; p2[0] = y; p3[0] = z;
        split3(v2, v3, x, y, z); p1[1] = x; p2[1] = y; p3[1] = z;
        split3(v4, v5, x, y, z); p1[2] = x; p2[2] = y; p3[2] = z;
        split3(v6, v7, x, y, z); p1[3] = x; p2[3] = y; p3[3] = z;
        dst[0] = p1; dst[NG_BLK] = p2; dst[2 * NG_BLK] = p3;
    };
    auto panel_step = [&](int p, f32x4 (&xa)[2][2], float (&xb)[2][8]) __attribute__((always_inline)) {
        __syncthreads();  // previous panel's images fully consumed
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            put(Aimg + adst[j], xa[j][0][0], xa[j][0][1], xa[j][0][2], xa[j][0][3], xa[j][1][0], xa[j][1][1], xa[j][1][2], xa[j][1][3]);
            put(Bimg + bdst[j], xb[j][0], xb[j][1], xb[j][2], xb[j][3], xb[j][4], xb[j][5], xb[j][6], xb[j][7]);
        }
        __syncthreads();
        fetch(p + 2, xa, xb);   // into the set just emptied (clamped: the last two refills re-read the last panel and are never used)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4* ap = Aimg + ((w * 2 + s2) * 3) * NG_BLK + h * NG_HB + c;
            const bf16x8 A1 = __builtin_bit_cast(bf16x8, ap[0]), A2 = __builtin_bit_cast(bf16x8, ap[NG_BLK]), A3 = __builtin_bit_cast(bf16x8, ap[2 * NG_BLK]);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                const u32x4* bp = Bimg + ((tl * 2 + s2) * 3) * NG_BLK + h * NG_HB + c;
                const bf16x8 B1 = __builtin_bit_cast(bf16x8, bp[0]), B2 = __builtin_bit_cast(bf16x8, bp[NG_BLK]), B3 = __builtin_bit_cast(bf16x8, bp[2 * NG_BLK]);
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, acc[tl], 0, 0, 0);
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, acc[tl], 0, 0, 0);
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, acc[tl], 0, 0, 0);
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, acc[tl], 0, 0, 0);
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, acc[tl], 0, 0, 0);
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[tl], 0, 0, 0);
            }
        }
    };
    fetch(0, pa[0], pb[0]);
    fetch(1, pa[1], pb[1]);
    int p = 0;
    for (; p + 1 < nb; p += 2) {
        panel_step(p, pa[0], pb[0]);
        panel_step(p + 1, pa[1], pb[1]);
    }
    if (p < nb) panel_step(p, pa[0], pb[0]);
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const int col = c0 + tl * 32 + c;
        if (col >= k) continue;
        const float sv = S ? S[col] : 1.0f;
        const float inv = sv > 0.0f ? 1.0f / sv : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < rows) out[(int64_t)row * ldo + col] = acc[tl][reg] * inv;
        }
    }
}

// sigma refinement of the tall path: Y = X Vr (unscaled) -> sigma_j = |y_j| (fp64, fixed order), u_j = y_j / sigma_j.
// |X v_j| is second-order accurate in the error of v_j and does not see the fp32 rounding of R.
__global__ __launch_bounds__(256) void colsumsq_kernel(TallBatch tb, int64_t ldy, int rows, int k, int rows_per_split, double* __restrict__ part_all,
                                                       int64_t part_stride) {
    const float* __restrict__ Y = tb.lng[blockIdx.z];
    if (!Y) return;
    double* __restrict__ part = part_all + (int64_t)blockIdx.z * part_stride;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, split = blockIdx.y;
    const int rb = split * rows_per_split, re = min(rb + rows_per_split, rows);
    double acc = 0.0;
    if (c < k)
        for (int r = rb + rl; r < re; r += 4) {
            const double v = Y[(int64_t)r * ldy + c];
            acc += v * v;
        }
    __shared__ double red[4][64];
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < k) part[(int64_t)split * k + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// one workgroup: ordered sum of the partials, sqrt, then a running minimum keeps S non-increasing (refined values of nearly equal
// singular values may swap by ~1e-6 relative; the columns are not re-ordered)
__global__ __launch_bounds__(256) void colfinish_kernel(TallBatch tb, const double* __restrict__ part_all, int64_t part_stride, int nsplit, int k,
                                                        float* __restrict__ inv_all) {
    if (!tb.lng[blockIdx.x]) return;
    const double* __restrict__ part = part_all + (int64_t)blockIdx.x * part_stride;
    float* __restrict__ S = tb.S[blockIdx.x];
    float* __restrict__ inv = inv_all + (int64_t)blockIdx.x * k;
    for (int c = threadIdx.x; c < k; c += 256) {
        double acc = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) acc += part[(int64_t)sp * k + c];
        const double sg = sqrt(acc);
        S[c] = (float)sg;
        inv[c] = sg > 0.0 ? (float)(1.0 / sg) : 0.0f;
    }
    __syncthreads();
    // running minimum, three short passes: per-thread chunk minima, exclusive prefix over the 256 chunk minima, apply
    __shared__ float cmin[256];
    const int chunk = (k + 255) / 256;
    const int c0 = threadIdx.x * chunk, c1 = min(c0 + chunk, k);
    float m = INFINITY;
    for (int c = c0; c < c1; ++c) m = fminf(m, S[c]);
    cmin[threadIdx.x] = m;
    __syncthreads();
    float run = INFINITY;
    for (int t = 0; t < (int)threadIdx.x; ++t) run = fminf(run, cmin[t]);
    for (int c = c0; c < c1; ++c) { run = fminf(run, S[c]); S[c] = run; }
}
__global__ void colscale_kernel(TallBatch tb, int64_t ldy, int rows, int k, const float* __restrict__ inv_all) {
    float* __restrict__ Y = tb.lng[blockIdx.z];
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r0 = blockIdx.y * 32;
    if (!Y || c >= k) return;
    const float* __restrict__ inv = inv_all + (int64_t)blockIdx.z * k;
    const float f = inv[c];
    for (int r = r0; r < min(r0 + 32, rows); ++r) Y[(int64_t)r * ldy + c] *= f;
}

