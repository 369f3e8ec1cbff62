// gram_i8.h — the fp64 Gram matrix of the Cholesky-QR reduction computed EXACTLY on the int8 matrix pipe (included by svd_jacobi.hip inside
// its anonymous namespace, after tall_kernels.h).  Round 6.
//
// gram64_kernel runs at what v_mfma_f64_16x16x4 delivers (49.7 TFLOP/s: 45 ms per 32 x 4096^2, the largest single item of the reduction).
// The Cholesky-QR needs G to ~1e-16 cond^2, which rules out every rounded product — but not an EXACT Gram matrix of a matrix that differs
// from X by a column-wise tiny perturbation (that is a backward error of the same kind as, and smaller than, the fp32 rounding of R that
// follows).  So (Ozaki-style error-free splitting, specialised to a Gram product):
//   1. every column j gets a power of two 2^E_j just above its largest magnitude (colmaxexp_kernel; max |x| 2^-E_j < 127/128);
//   2. t = rint(x 2^(23 - E_j)) is a 24-bit signed integer, written as three balanced radix-256 digits t = d0 2^16 + d1 2^8 + d2 with
//      d0 in [-127, 127], d1, d2 in [-128, 127] (split_i8_kernel: three int8 planes in MFMA operand order) — entries within a factor 2 of the
//      column's largest are kept exactly, the others to 2^-24 of it: |x~ - x| <= 2^(E_j - 24), about 2e-7 of the column norm for a
//      Gaussian-like column (fp32 itself: 6e-8);
//   3. gram_i8_kernel accumulates the nine digit products D_a^T D_b with v_mfma_i32_32x32x32_i8 into FIVE int32 accumulators, one per
//      weight s = a + b (at most 3 x 2^14 per row and accumulator: exact in int32 for <= 32768 rows; longer problems go in row segments that
//      are added in fp64), and combines them as sum_s P_s 2^(8 (4 - s)) 2^(E_i + E_j - 46) in fp64: the exact Gram matrix of X~ up to the
//      2^-53 of that last sum.
// 9 x 2 x 4096^3 / 2 integer operations per 4096^2 problem at ~2 Pop/s instead of 6.9e10 fp64 flop at 49.7 TFLOP/s.
// Inf / NaN in a column: its exponent is a sentinel and its row and column of G come out NaN, as gram64_kernel's do.
#pragma once

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int GI_BAD = 0x7fffffff;   // exponent sentinel: the column holds Inf / NaN
constexpr int GI_ZERO = -0x40000000; // exponent of an all-zero column (digits are all zero, the scale does not matter)

// ex[b][col] = E with  max_r |X[r][col]| 2^-E < 127/128;  dn[b][col] (optional) = |x_col|_2 in fp64, fixed summation order
// (grid: (nb, batch), 256 threads: lane = column, eight row lanes)
__global__ __launch_bounds__(256) void colmaxexp_kernel(const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int m_pad,
                                                        int n_pad, int* __restrict__ ex, double* __restrict__ dn) {
    __shared__ unsigned smax[8][PB];
    __shared__ double ssq[8][PB];
    const int P = blockIdx.x, b = blockIdx.y, c = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float* __restrict__ xp = X + (int64_t)b * batch_stride + (int64_t)P * panel_stride + c;
    unsigned mx = 0;
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    for (int r = rl; r < m_pad; r += 32) {  // m_pad is a multiple of 32: four independent loads per trip
        const float x0 = xp[(int64_t)r * PB], x1 = xp[(int64_t)(r + 8) * PB], x2 = xp[(int64_t)(r + 16) * PB], x3 = xp[(int64_t)(r + 24) * PB];
        const unsigned a0 = __float_as_uint(x0) & 0x7fffffffu, a1 = __float_as_uint(x1) & 0x7fffffffu;
        const unsigned a2 = __float_as_uint(x2) & 0x7fffffffu, a3 = __float_as_uint(x3) & 0x7fffffffu;
        const unsigned m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3, m4 = m01 > m23 ? m01 : m23;
        mx = mx > m4 ? mx : m4;  // magnitudes of finite floats order like their bit patterns; Inf / NaN patterns are above all of them
        q0 = fma((double)x0, (double)x0, q0); q1 = fma((double)x1, (double)x1, q1);
        q2 = fma((double)x2, (double)x2, q2); q3 = fma((double)x3, (double)x3, q3);
    }
    smax[rl][c] = mx;
    ssq[rl][c] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (rl == 0) {
        double q = ssq[0][c];
#pragma unroll
        for (int i = 1; i < 8; ++i) { mx = mx > smax[i][c] ? mx : smax[i][c]; q += ssq[i][c]; }
        int E;
        if (mx >= 0x7f800000u) E = GI_BAD;
        else if (mx == 0u) E = GI_ZERO;
        else {
            int e2;
            const float f = frexpf(__uint_as_float(mx), &e2);  // mx = f 2^e2, f in [0.5, 1)
            E = e2 + (f >= 127.0f / 128.0f ? 1 : 0);
        }
        ex[(int64_t)b * n_pad + P * PB + c] = E;
        if (dn) dn[(int64_t)b * n_pad + P * PB + c] = sqrt(q);
    }
}

// exs[b][i] = ex[b][perm[b][i]], dp[b][i] = d[b][perm[b][i]]: exponents and norms in sorted-column order; inv[b][perm[b][i]] = i
__global__ void perm_gather_kernel(const int* __restrict__ ex, const double* __restrict__ d, const int* __restrict__ perm, int n, int* __restrict__ exs,
                                   double* __restrict__ dp, int* __restrict__ inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int q = perm[(int64_t)b * n + i];
    exs[(int64_t)b * n + i] = ex[(int64_t)b * n + q];
    dp[(int64_t)b * n + i] = d[(int64_t)b * n + q];
    inv[(int64_t)b * n + q] = i;
}

// Digit planes of the rows [r0, r0 + 16 kgs) of every problem:  planes[b][digit a][panel P][16-row group kg][column c][16 bytes = rows].
// One (P, kg) piece of a digit is 512 contiguous bytes; two consecutive pieces are the A (or B) operand of one v_mfma_i32_32x32x32_i8 of a
// wave: lane l = 32 (kg & 1) + c reads its 16 bytes at offset 16 l.  Rows >= m_pad are zero.
// grid: (nb, ceil(kgs / 8), batch), 256 threads = 8 row groups x 32 columns.  The workgroup READS panel P of X (coalesced) and writes column q = 32 P + c
// to plane position inv[q] (nullable: q itself) — a scatter of 16-byte pieces.  (First version: plane panel P GATHERED its columns perm[32 P + c], every lane
// of a load in another panel's 128-byte row: 5.6 ms per 32 x 4096^2 against 0.7 ms unpermuted.)
__global__ __launch_bounds__(256) void split_i8_kernel(const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int nb, int m_pad,
                                                       int n_pad, const int* __restrict__ ex /* exponents of the columns of X, in X's order */, int r0, int kgs,
                                                       signed char* __restrict__ planes, int64_t plane_stride /* bytes per digit and problem = nb kgs 512 */,
                                                       const int* __restrict__ inv /* nullable: plane position of every column of X */,
                                                       const int* __restrict__ done /* nullable: problems whose flag is set are skipped */) {
    const int P = blockIdx.x, b = blockIdx.z, c = threadIdx.x & 31;
    const int kg = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (kg >= kgs || (done && ld_flag(done + b))) return;
    const int q = P * PB + c;
    const int E = ex[(int64_t)b * n_pad + q];
    const int pos = inv ? inv[(int64_t)b * n_pad + q] : q;
    const float* __restrict__ xp = X + (int64_t)b * batch_stride + (int64_t)P * panel_stride + c;
    unsigned w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
    if (E != GI_BAD && E != GI_ZERO) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + kg * 16 + i;
            const float x = r < m_pad ? xp[(int64_t)r * PB] : 0.0f;
            const int t = (int)rintf(ldexpf(x, 23 - E));       // exact scaling, |t| <= 127 * 2^16
            const int d2 = (int)(signed char)(t & 0xff);
            const int t1 = (t - d2) >> 8;
            const int d1 = (int)(signed char)(t1 & 0xff);
            const int d0 = (t1 - d1) >> 8;
            w0[i >> 2] |= (unsigned)(d0 & 0xff) << (8 * (i & 3));
            w1[i >> 2] |= (unsigned)(d1 & 0xff) << (8 * (i & 3));
            w2[i >> 2] |= (unsigned)(d2 & 0xff) << (8 * (i & 3));
        }
    }
    signed char* pb = planes + (int64_t)b * 3 * plane_stride + ((int64_t)(pos >> 5) * kgs + kg) * 512 + (pos & 31) * 16;
    *(uint4*)(pb) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
    *(uint4*)(pb + plane_stride) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    *(uint4*)(pb + 2 * plane_stride) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
}

// G[b] (upper 32-blocks, I <= J) (+)= exact Gram matrix of the digitised rows of this segment.
// Workgroup = 512 threads = one 128 x 128 block (Ib <= Jb) of one problem; wave w: I panels 4 Ib + 2 (w >> 2) + {0, 1}, J panel 4 Jb + (w & 3)
// (64 x 32 of the block: 2 tiles x 5 weights x 16 = 160 accumulator registers, two waves per SIMD).  A stage = 64 rows of the four I panels and the
// four J panels, three digits each: 2 x 24 KB through LDS in the planes' own order (linear 16-byte pieces: no bank conflicts either way), double
// buffered; per stage and wave 2 x 18 matrix instructions on 2 x 9 operand reads.
constexpr int GI_STAGE_BYTES = 2 * 3 * 4 * 4 * 512;  // side, digit, panel, row group
__global__ __launch_bounds__(512) void gram_i8_kernel(const signed char* __restrict__ planes, int64_t plane_stride, int nb, int kgs, int n_pad,
                                                      const int* __restrict__ ex, double* __restrict__ G, int64_t ldg, int64_t g_batch_stride,
                                                      int accumulate, int nt, const double* __restrict__ dp /* nullable: G_ij / (dp_i dp_j) is stored;
                                                      a column with dp = 0 gets a unit diagonal and zeros */, int order /* block order: 0 row runs per XCD, 1 4 x 8 groups per XCD, 2 plain */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gi_lds[];  // 2 x GI_STAGE_BYTES
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD a contiguous run of blocks (they share I panels in its L2)
    // (a block count that is not a multiple of 8 keeps the plain order: correct, merely less local)
    const int total = gridDim.x;
    const int lid = ((total & 7) || order == 2) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3));
    int Ib = 0, Jb = 0;
    if (order == 1) {
        // 4 x 8 groups of blocks (the 32 workgroups an XCD runs at a time share 4 I-side and 8 J-side column blocks in its L2), groups row by row
        int rem = lid;
        bool found = false;
        for (int SI = 0; SI * 4 < nt && !found; ++SI)
            for (int SJ = (SI * 4) >> 3; SJ * 8 < nt && !found; ++SJ) {
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ib = SI * 4 + i, lo = max(SJ * 8, ib), hi = min(SJ * 8 + 8, nt);
                    if (ib < nt && hi > lo) cnt += hi - lo;
                }
                if (rem >= cnt) { rem -= cnt; continue; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ib = SI * 4 + i, lo = max(SJ * 8, ib), hi = min(SJ * 8 + 8, nt);
                    const int wdt = (ib < nt && hi > lo) ? hi - lo : 0;
                    if (!found && rem < wdt) { Ib = ib; Jb = lo + rem; found = true; }
                    if (!found) rem -= wdt;
                }
            }
    } else {
        int rem = lid;
        while (rem >= nt - Ib) { rem -= nt - Ib; ++Ib; }
        Jb = Ib + rem;
    }
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w >> 2, wj = w & 3;
    const signed char* __restrict__ pl = planes + (int64_t)b * 3 * plane_stride;

    // loader: piece q of a stage (16 bytes), q = round * 512 + tid, in LDS order [side][digit][panel][row group][column]
    const unsigned char* gsrc[6];
    bool gok[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int q = r * 512 + tid;
        const int side = q / 1536, qq = q % 1536;
        const int a = (qq >> 7) >> 2, p4 = (qq >> 7) & 3, within = qq & 127;  // within: row group (2 bits) x column (5 bits)
        const int P = 4 * (side ? Jb : Ib) + p4;
        gok[r] = P < nb;
        gsrc[r] = (const unsigned char*)pl + (int64_t)a * plane_stride + (int64_t)(gok[r] ? P : 0) * kgs * 512 + within * 16;
    }
    i32x16 acc[2][5];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][s][i] = 0;

    const int nstage = kgs >> 2;  // kgs is a multiple of 4 (64-row stages)
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
                        acc[t][a + c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[t][a], fb[c], acc[t][a + c], 0, 0, 0);
        }
        if (st + 1 < nstage) {
            unsigned char* nxt = gi_lds + ((st + 1) & 1) * GI_STAGE_BYTES;
#pragma unroll
            for (int r = 0; r < 6; ++r) *(uint4*)(nxt + (r * 512 + tid) * 16) = stg[r];
        }
        __syncthreads();
    }
    // D[i][j]: j = lane & 31 (B operand = J panel), i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A operand = I panel)
    const int J = 4 * Jb + wj;
    if (J >= nb) return;
    const int j = lane & 31;
    const int Ej = ex[(int64_t)b * n_pad + J * PB + j];
    double sj = Ej == GI_BAD ? __builtin_nan("") : (Ej == GI_ZERO ? 0.0 : ldexp(1.0, Ej - 23));
    if (dp) { const double dj = dp[(int64_t)b * n_pad + J * PB + j]; sj = dj > 0.0 ? sj / dj : (dj == 0.0 ? 0.0 : __builtin_nan("")); }
    double* __restrict__ out = G + (int64_t)b * g_batch_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int I = 4 * Ib + 2 * wi + t;
        if (I > J || I >= nb) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const int Ei = ex[(int64_t)b * n_pad + I * PB + i];
            double si = Ei == GI_BAD ? __builtin_nan("") : (Ei == GI_ZERO ? 0.0 : ldexp(1.0, Ei - 23));
            bool unit = false;
            if (dp) {
                const double di = dp[(int64_t)b * n_pad + I * PB + i];
                si = di > 0.0 ? si / di : (di == 0.0 ? 0.0 : __builtin_nan(""));
                unit = di == 0.0 && I == J && i == j;
            }
            double v = (double)acc[t][4][reg];
            v += (double)acc[t][3][reg] * 256.0;
            v += (double)acc[t][2][reg] * 65536.0;
            v += (double)acc[t][1][reg] * 16777216.0;
            v += (double)acc[t][0][reg] * 4294967296.0;
            v = v * si * sj;
            const int64_t o = (int64_t)(I * PB + i) * ldg + J * PB + j;
            v = accumulate ? out[o] + v : v;
            out[o] = unit ? 1.0 : v;
        }
    }
}
