// twolevel.h — kernels of the TWO-LEVEL dense sweep (included by svd_jacobi.hip inside its anonymous namespace).
//
// Why: at panel width 32 a sweep moves (nb-1) x 3 panel passes through HBM and sits on the roofline ridge.  The
// two-level sweep works on SUPER-PANELS of 64 columns (two adjacent 32-column panels, layout unchanged) under the same XOR
// schedule, now over ns = nb/2 super-panels.  The four 32-blocks of a super-pair (S, T) are S0, S1, T0, T1 = 0..3.  Per step:
//   sgram6   ONE pass over the four panels: the four cross tiles [0,2] [0,3] [1,2] [1,3] and the two within tiles [0,1] [2,3]
//            (fp32 MFMA, row-split partials).  The 32x32 DIAGONAL blocks of the panels are CARRIED: every 64x64 eigen-solve leaves
//            Q^T G Q behind, whose diagonal blocks are the Gram blocks of the two updated panels; they are refreshed from the data
//            once per sweep (internal step, below).  Runs only in front of the first super-step of a sweep (and wherever the fused
//            kernel below does not apply): supgram leaves the tiles of the next step behind.
//   evdw12   (evd_wave.hip) both inner steps of every super-pair in one launch, one wave per 64x64 solve.  Inner step 0: sub-pairs
//            (0,2) and (1,3), assembled from the carried blocks and the summed cross tiles; inner step 1: sub-pairs (0,3) and (1,2), whose
//            cross blocks are the tile G[{0,2},{1,3}] transformed by the two Q's of step 0 (two small fp32-MFMA products); the epilogue emits
//            the sub-pair's 128x64 column block of Qfin = Q^(0) Q^(1), the new carried diagonal blocks, and the squared column norms the
//            pair had BEFORE its rotation (EvdV3::Din: the column scales of supgram's split-fp16 arithmetic).  No 128x128 Gram matrix is
//            ever formed.
//   supgram  ONE pass: [X_S X_T] <- [X_S X_T] Qfin (128x128, split-fp16 on the fp16 matrix pipe) AND the six tiles of the next step's pairs
//   supdate_split   the update alone (split-bf16): last super-step of a sweep, padded schedules, and the fallback of a call that turned NaN
//            on the split-fp16 path
// The pairs INSIDE a super-panel (2S, 2S+1) are the d = 1 step of the single-level schedule: it runs first in every sweep with the
// single-level kernels (full 3-block Gram from the data) and its eigen-solve emits the fresh carried blocks of both panels.
// HBM passes per sweep: 2 (nb/2 - 1) + 4 instead of 3 (nb - 1); launches per super-step: 2; the number of 64x64 eigen-solves is
// unchanged (every 32-panel pair still meets exactly once per sweep).  Each eigen-solve sorts its own 64 columns (larger half to
// the lower panel); there is no global 128-column sort (CPU prototype tools/proto_two_level.py: same sweep count either way).

constexpr int SW = 64;       // super-panel width
constexpr int SP = 2 * SW;   // super-pair width

// a super-pair is updated when any of its four eigen-solves rotated
__device__ __forceinline__ bool pair_active(const int* __restrict__ subact, int64_t slot) {
    const int* sa = subact + slot * 4;
    return (ld_flag(sa) | ld_flag(sa + 1) | ld_flag(sa + 2) | ld_flag(sa + 3)) != 0;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // {hi16(x1) : hi16(x0)}
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);  // exact
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);  // exact, <= 8 bits left
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}


// --------------------------------------------------------------------------------------------------
// sgram6: partial Gram tiles of a super-pair over a row range: Gx6[split][t] (32x32 row-major), t = 0..5 -> [0,2] [0,3] [1,2] [1,3]
// [0,1] [2,3], tile [x,y] = X_x[rows]^T X_y[rows].  Same streaming structure as gram_kernel (register prefetch of the next 16-row
// chunk, wave-private LDS image, one ds_read_b32 per MFMA operand); four panels and six accumulators per wave.
constexpr int SGRAM6_SMEM_FLOATS = 4 * 4 * 16 * PB;  // 32 KiB
__global__ __launch_bounds__(256, 2) void sgram6_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int ns, int D,
                                                        int m_pad, int rows_per_split, float* __restrict__ Gx, const int* __restrict__ done) {
    __shared__ __attribute__((aligned(16))) float smem[SGRAM6_SMEM_FLOATS];
    const int split = blockIdx.x, pair = blockIdx.y, b = blockIdx.z;
    const int nsplit = gridDim.x, npairs = gridDim.y;
    if (ld_flag(done + b)) return;
    int S, T;
    super_pair(sc, ns, D - 1, pair, S, T);
    if (T >= ns) return;
    const float* __restrict__ Xb = X + (int64_t)b * batch_stride;
    const float* __restrict__ P0 = Xb + (int64_t)(2 * S) * panel_stride;
    const float* __restrict__ P1 = Xb + (int64_t)(2 * S + 1) * panel_stride;
    const float* __restrict__ P2 = Xb + (int64_t)(2 * T) * panel_stride;
    const float* __restrict__ P3 = Xb + (int64_t)(2 * T + 1) * panel_stride;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r_begin = split * rows_per_split;
    const int r_end = min(r_begin + rows_per_split, m_pad);
    constexpr int SCH = 16;  // rows per staged chunk: 8 KiB per wave, 32 KiB per workgroup
    const int nchunks = (r_end - r_begin) / SCH;  // m_pad and rows_per_split are multiples of 32

    float (*stage)[4 * SCH * PB] = (float (*)[4 * SCH * PB])smem;  // per wave: 16 rows of the four panels
    float* s = stage[w];
    f32x16 a02 = {0}, a03 = {0}, a12 = {0}, a13 = {0}, a01 = {0}, a23 = {0};
    {
        f32x4 p0[SCH / 8], p1[SCH / 8], p2[SCH / 8], p3[SCH / 8];
        auto fetch = [&](int ch) {
            const int64_t r0 = r_begin + (int64_t)ch * SCH;
#pragma unroll
            for (int it = 0; it < SCH / 8; ++it) {
                const int64_t o = (r0 + it * 8) * PB + lane * 4;
                p0[it] = *(const f32x4*)(P0 + o);
                p1[it] = *(const f32x4*)(P1 + o);
                p2[it] = *(const f32x4*)(P2 + o);
                p3[it] = *(const f32x4*)(P3 + o);
            }
        };
        if (w < nchunks) fetch(w);
        for (int ch = w; ch < nchunks; ch += 4) {
#pragma unroll
            for (int it = 0; it < SCH / 8; ++it) {
                *(f32x4*)(s + 0 * (SCH * PB) + it * 256 + lane * 4) = p0[it];
                *(f32x4*)(s + 1 * (SCH * PB) + it * 256 + lane * 4) = p1[it];
                *(f32x4*)(s + 2 * (SCH * PB) + it * 256 + lane * 4) = p2[it];
                *(f32x4*)(s + 3 * (SCH * PB) + it * 256 + lane * 4) = p3[it];
            }
            if (ch + 4 < nchunks) fetch(ch + 4);
            // fp32 MFMA (the split-bf16 form was measured in round 2: this pass is latency / HBM-, not matrix-pipe-bound, and it runs once per sweep)
#pragma unroll
            for (int u = 0; u < SCH / 2; ++u) {
                const float x0 = s[0 * (SCH * PB) + u * 64 + lane], x1 = s[1 * (SCH * PB) + u * 64 + lane];
                const float y0 = s[2 * (SCH * PB) + u * 64 + lane], y1 = s[3 * (SCH * PB) + u * 64 + lane];
                a02 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a02, 0, 0, 0);
                a03 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a03, 0, 0, 0);
                a12 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a12, 0, 0, 0);
                a13 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a13, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, a01, 0, 0, 0);
                a23 = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, y1, a23, 0, 0, 0);
            }
        }
    }
    // cross-wave reduction in a fixed order ((w0 + w2) + (w1 + w3)) through the staging memory, two tiles per round
    // (4 waves x 2 tiles x 4 KiB = 32 KiB), coalesced store
    float* red = &stage[0][0];
    float* __restrict__ out = Gx + (((int64_t)b * npairs + pair) * nsplit + split) * (6 * 1024);
#pragma unroll
    for (int rnd = 0; rnd < 3; ++rnd) {
        __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            red[w * 2048 + (0 * 16 + reg) * 64 + lane] = rnd == 0 ? a02[reg] : (rnd == 1 ? a12[reg] : a01[reg]);
            red[w * 2048 + (1 * 16 + reg) * 64 + lane] = rnd == 0 ? a03[reg] : (rnd == 1 ? a13[reg] : a23[reg]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = tid + 256 * q;
            const float v = (red[p] + red[2 * 2048 + p]) + (red[2048 + p] + red[3 * 2048 + p]);
            const int rg = p >> 6, ln = p & 63;
            const int tile = 2 * rnd + (rg >> 4), reg = rg & 15;
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), j = ln & 31;
            out[tile * 1024 + i * 32 + j] = v;
        }
    }
}

constexpr int ULD = SP + 4;  // LDS row stride in floats (528 B: conflict-free b128 row-per-lane reads)

// --------------------------------------------------------------------------------------------------
// supdate with split-bf16 arithmetic.  Every fp32 value is written EXACTLY as the sum of three bf16 numbers (truncation split:
// 8 + 8 + 8 significant bits), x = x1 + x2 + x3, and the product X Q is evaluated on the bf16 matrix pipe as
//   X1 Q1 + (X1 Q2 + X2 Q1) + (X2 Q2 + X1 Q3 + X3 Q1)           (the dropped terms are <= 2^-24 |x| |q|: fp32 rounding level)
// with fp32 accumulation: 6 x 8 = 48 v_mfma_f32_32x32x16_bf16 (32 cycles each) per 32x128 tile and wave instead of 64
// v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x less matrix-pipe time for the kernel that holds 4/5 of the sweep's flops.
// Bitwise it is not the fp32 MFMA result (different summation tree), numerically it is equivalent (tests compare both with fp64).
constexpr int SUPDATE_SMEM_FLOATS = 2 * 32 * ULD;  // 33 KiB
__global__ __launch_bounds__(256, 2) void supdate_split_kernel(Sched sc, float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int ns,
                                                               int D, int R, int rows_per_wg, const float* __restrict__ Qfin,
                                                               const int* __restrict__ subact, const int* __restrict__ done, int* __restrict__ nupd) {
    __shared__ __attribute__((aligned(16))) float smem[SUPDATE_SMEM_FLOATS];
    const int chunk = blockIdx.x, pair = blockIdx.y, b = blockIdx.z, npairs = gridDim.y;
    if (ld_flag(done + b) || !pair_active(subact, (int64_t)b * npairs + pair)) return;
    int S, T;
    super_pair(sc, ns, D - 1, pair, S, T);
    if (T >= ns) return;
    if (chunk == 0 && threadIdx.x == 0) atomicAdd(&nupd[b], 1);  // instrumentation: super-pairs updated in this sweep
    float* __restrict__ Xb = X + (int64_t)b * batch_stride;
    float* __restrict__ P0 = Xb + (int64_t)(2 * S) * panel_stride;
    float* __restrict__ P1 = Xb + (int64_t)(2 * S + 1) * panel_stride;
    float* __restrict__ P2 = Xb + (int64_t)(2 * T) * panel_stride;
    float* __restrict__ P3 = Xb + (int64_t)(2 * T + 1) * panel_stride;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, c = lane & 31;
    const float* __restrict__ Qp = Qfin + ((int64_t)b * npairs + pair) * (SP * SP);
    // B operand of k-step s: lane (j = c, group h) holds Q[16 s + 8 h + e][32 w + c], e = 0..7, in three bf16 parts
    u32x4 q1[8], q2[8], q3[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Qp[(16 * s + 8 * h + e) * SP + 32 * w + c];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            unsigned a, bb, cc;
            split3(v[2 * e2], v[2 * e2 + 1], a, bb, cc);
            q1[s][e2] = a; q2[s][e2] = bb; q3[s][e2] = cc;
        }
    }
    float* __restrict__ Pw = (w == 0) ? P0 : (w == 1) ? P1 : (w == 2) ? P2 : P3;

    float (*tile)[32 * ULD] = (float (*)[32 * ULD])smem;
    const int r_begin = chunk * rows_per_wg;
    const int r_end = min(r_begin + rows_per_wg, R);
    if (r_begin >= r_end) return;
    f32x4 pre0, pre1, pre2, pre3;
    auto fetch = [&](int r0) {
        const int64_t o = (int64_t)r0 * PB + tid * 4;
        pre0 = *(const f32x4*)(P0 + o);
        pre1 = *(const f32x4*)(P1 + o);
        pre2 = *(const f32x4*)(P2 + o);
        pre3 = *(const f32x4*)(P3 + o);
    };
    auto stash = [&](float* t) {
        float* dst = t + (tid >> 3) * ULD + (tid & 7) * 4;
        *(f32x4*)(dst + 0) = pre0;
        *(f32x4*)(dst + 32) = pre1;
        *(f32x4*)(dst + 64) = pre2;
        *(f32x4*)(dst + 96) = pre3;
    };
    int cur = 0;
    fetch(r_begin);
    stash(tile[0]);
    __syncthreads();
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
        const bool more = r0 + 32 < r_end;
        if (more) fetch(r0 + 32);
        const float* my = tile[cur] + c * ULD + 8 * h;
        f32x16 acc = {0};  // one accumulator: the small terms of a k-step go in first
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 v0 = *(const f32x4*)(my + 16 * s);
            const f32x4 v1 = *(const f32x4*)(my + 16 * s + 4);
            u32x4 a1, a2, a3;
            {
                unsigned x, y, z;
                split3(v0[0], v0[1], x, y, z); a1[0] = x; a2[0] = y; a3[0] = z;
                split3(v0[2], v0[3], x, y, z); a1[1] = x; a2[1] = y; a3[1] = z;
                split3(v1[0], v1[1], x, y, z); a1[2] = x; a2[2] = y; a3[2] = z;
                split3(v1[2], v1[3], x, y, z); a1[3] = x; a2[3] = y; a3[3] = z;
            }
            const bf16x8 A1 = __builtin_bit_cast(bf16x8, a1), A2 = __builtin_bit_cast(bf16x8, a2), A3 = __builtin_bit_cast(bf16x8, a3);
            const bf16x8 B1 = __builtin_bit_cast(bf16x8, q1[s]), B2 = __builtin_bit_cast(bf16x8, q2[s]), B3 = __builtin_bit_cast(bf16x8, q3[s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            Pw[(int64_t)(r0 + i) * PB + c] = acc[reg];
        }
        if (more) stash(tile[cur ^ 1]);
        __syncthreads();
        cur ^= 1;
    }
}

// --------------------------------------------------------------------------------------------------
// supgram: the update of super-step D fused with the Gram tiles of super-step E (the step that follows).  Under the XOR ordering the
// four super-panels {a, a^D, a^E, a^D^E} (a QUAD) are closed under both steps: the pairs (a, a^D) and (a^E, a^D^E) rotate now, the
// pairs (a, a^E) and (a^D, a^D^E) meet next.  One workgroup of EIGHT waves streams a row chunk of a quad ONCE, in 32-row tiles; wave w
// owns output panel w (waves 0-3: first pair, 4-7: second pair; its 128x32 slice of Qfin sits pre-split in 64 VGPRs), writes it back,
// and leaves it — already split, in MFMA operand order — in LDS.  The C layout of the update (lane = column, registers = rows) IS the
// A/B operand layout of the Gram product over rows (the reduction index may be permuted freely as long as both operands agree), so the
// twelve 32x32 tiles of the two next pairs are accumulated from those images without any transpose: 24 half-tiles (tile x 16-row
// k-step), three per wave, held in registers over the whole chunk.  HBM traffic of a super-step: one read + one write instead of
// read + (read + write).  Super-panels beyond ns (power-of-two padding of the schedule) read as absent: no loads, no stores.
//
// PING-PONG SCHEDULE (round 4).  Rounds 2-3 ran all eight waves through the same phases between two barriers per tile — fetch / split /
// LDS stores, then the matrix instructions, then panel stores — so the matrix pipe, the VALU and the memory pipe took turns: the kernel
// ran at the SUM of its HBM time and its MFMA time.  The two waves of a SIMD are w and w + 4, i.e. one wave of each pair.  Now the pairs
// run HALF A TILE APART: between two barriers the waves of one pair are in their COMPUTE segment (Gram MFMAs of an earlier tile + the
// update MFMAs of this one, operands from LDS) while the waves of the other pair are in their MEMORY segment (split + LDS stores of the
// next incoming tile, global loads of the tile after that, panel stores of the tile just computed, its accumulators as Gram operands),
// then they swap: every SIMD holds one wave that feeds the matrix pipe and one that feeds VALU / LDS / HBM at any time.  Every pair
// stages only ITS OWN 32 x 128 incoming tile — written in its memory segment, read in its compute segment — so the incoming image needs
// no double buffer; the Gram operand image is double buffered instead, because the tiles of the next step need the updated panels of
// BOTH pairs and pair B's lag half a tile behind: pair A accumulates the Gram units of tile t - 2 next to the update of tile t, pair B
// those of tile t - 1 (timeline in segments: A computes tile t in segment 2t and publishes its operands in 2t + 1, B computes in 2t + 1
// and publishes in 2t + 2; the operand buffer (t & 1) is complete after segment 2t + 2, read in 2t + 3 (B) and 2t + 4 (A), and rewritten
// from 2t + 5).  Two barriers per tile; global loads and stores stay in flight across them (s_barrier waits for lgkmcnt only).
//
// SPLIT-FP16 ARITHMETIC WITH POWER-OF-TWO COLUMN SCALES (round 4).  Telemetry (tools/power_probe.py) shows this kernel pinned at the
// 1400 W socket cap at 1.65 GHz with the matrix pipe half busy: its time is its ENERGY, and two thirds of that are the matrix
// instructions.  Rounds 2-3 wrote every fp32 value as three bf16 numbers (8 + 8 + 8 bits) and every fp32 product as six bf16 products.
// fp16 carries 11 bits: x = h1 + h2 to 22 bits and x q = h1 q1 + h1 q2 + h2 q1 to 2^-22 — THREE products, half the matrix work and two
// thirds of the LDS traffic — but only where both parts stay inside fp16's range, 2^-24 .. 65504, which an unscaled column of a graded
// matrix does not (the round-3 trial without scales stalled at 1e-3 orthogonality).  So every column is used at a power-of-two scale
// (exact):   X Q = (X 2^-ein) (2^ein Q 2^-eout) 2^eout,
//   ein[k]   from the column's squared norm before the rotation (v3.Din, written by the eigen-solve launch of this step): the scaled
//            column has norm <= 2^9, so its entries are <= 2^9 and the accumulators stay below 2^12.5 — inside fp16 when they become the
//            Gram operands, with 3 bits to spare if the carried norm is off; columns more than 2^20 below the largest column of their pair
//            share its floor (noise columns of rank-deficient inputs: their carried norms mean nothing);
//   eout[j]  from the largest TERM of output column j, max_k |Q[k][j]| 2^ein[k] (not from its norm: a column that cancels down to noise
//            has terms far above its norm, and fp32 itself resolves it only relative to those terms): the scaled Q has entries <= 1.
// Measured against fp64 (tools/emu, tests/test_gpu_twolevel.py): 3.0e-7 per column against 2.2e-7 for a plain fp32 product and 1.1e-7
// for the six-product bf16 form.  The Gram tiles come out at the scale 2^(eout[i] + eout[j]) and are rescaled when they are stored.
// A pair at rest (or one with an absent member) goes through the same arithmetic with Q = I and is not written back.
// If a call ever produces a NaN through this path (a carried norm that is wrong by more than 2^7), the driver repeats it with the
// separate passes (sgram6 in fp32, supdate_split in split-bf16), which need no scales.
//
// The incoming tile is split ONCE, by the thread that fetched it, and stored as the A operands of the update: image [pair][k-step][part]
// [lane] of 16-byte operands.  A 32-lane half block is padded to 36 operands and a (k-step, part) block to 73: the stash writes of an
// 8-lane group — two rows x four (k-step, lane group) targets — then fall on eight distinct 16-byte bank groups.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
constexpr int SG_HB = 36, SG_BLK = 73;
constexpr int SUPGRAM_AIMG_WORDS = 2 * 8 * 2 * SG_BLK * 4;         // one 32-row image of the eight panels, as fp16 x 2 A operands (36.5 KiB)
constexpr int SUPGRAM_OPND_WORDS = 8 * 2 * 2 * 64 * 4;             // updated panels as Gram operands [panel][k-step][part][lane] x 16 B (32 KiB)
constexpr int SUPGRAM_RED_FLOATS = 24 * 1024;                      // the final reduction reuses the front of the buffer
constexpr int SUPGRAM_MAIN_FLOATS = SUPGRAM_AIMG_WORDS + 2 * SUPGRAM_OPND_WORDS > SUPGRAM_RED_FLOATS ? SUPGRAM_AIMG_WORDS + 2 * SUPGRAM_OPND_WORDS : SUPGRAM_RED_FLOATS;
constexpr int SUPGRAM_SMEM_FLOATS = SUPGRAM_MAIN_FLOATS + 256;  // + exponent table: 103,936 B
constexpr int SG_TARGET = 9;    // scaled columns have norm <= 2^SG_TARGET
constexpr int SG_SPAN = 20;     // columns more than 2^SG_SPAN below the largest of their pair share its floor

// exponent e with sqrt(d) <= 2^e (d a squared norm); d <= 0, denormal or NaN -> very small
__device__ __forceinline__ int sg_half_exp(float d) {
    const int bits = __float_as_int(d);
    const int ex = (bits >> 23) & 255;
    if (bits <= 0 || ex == 0 || ex == 255) return -200;
    return (ex - 126 + 1) >> 1;   // d = f 2^p, f in [0.5, 1), p = ex - 126: ceil(p / 2)
}
// two fp32 -> packed (h1, h2) fp16 pairs with x ~ h1 + h2 (22 bits).  Written out as the instructions they should be: left to the compiler a pair
// cost 7 VALU instructions (it rounds h1 twice: once packed with v_cvt_pk_f16_f32 behind two v_mul, once per half with v_fma_mixlo_f16 to feed the
// residual); here 4 with a scale (3 without): v_fma_mix{lo,hi}_f16 evaluate fma(x, m, -h1) in fp32 from mixed fp32 / fp16 sources and round once.
//   split2s: h1 = fp16(x m), h2 = fp16(x m - h1), m a power of two (x m exact)          split2: the same with m = 1
__device__ __forceinline__ void split2s(float x0, float x1, float m0, float m1, unsigned& p1, unsigned& p2) {
    unsigned h1, h2;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(x0), "v"(m0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(x1), "v"(m1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(h2) : "v"(x0), "v"(m0), "v"(h1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h2) : "v"(x1), "v"(m1), "v"(h1));
    p1 = h1;
    p2 = h2;
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2) {
    unsigned h1, h2;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(h2) : "v"(x0), "v"(h1));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h2) : "v"(x1), "v"(h1));
    p1 = h1;
    p2 = h2;
}

#ifdef ASVD_SG_TIMING   // tools/bench_supgram.py --timing: shader cycles per stage of the tile loop, summed in scalar registers (no memory traffic in
                        // the loop), one wave per pair of workgroup (0, 0, 0); stage k = from stamp k - 1 to stamp k in program order
__device__ unsigned long long g_sg_ts[2][10];
#define SG_TS(i) do { __builtin_amdgcn_sched_barrier(0); { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sg_acc[i] += now_ - sg_last; sg_last = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define SG_ABL(bit) (sc.meas & (bit))   // timing-only ablations (results wrong): 1 no panel stores, 2 no fetch, 4 no matrix instructions, 8 no LDS operand stores
#else
#define SG_TS(i) do { } while (0)
#define SG_ABL(bit) 0
#endif
__global__ __launch_bounds__(512, 1) void supgram_kernel(Sched sc, float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int ns, int D,
                                                         int E, int R, int m_pad, int rows_per_wg, const float* __restrict__ Qfin,
                                                         const int* __restrict__ subact, const float* __restrict__ Din, float* __restrict__ Gx,
                                                         const int* __restrict__ done, int* __restrict__ nupd, int npairs) {
    extern __shared__ __attribute__((aligned(16))) float sg_smem[];
    const int chunk = blockIdx.x, quad = blockIdx.y, b = blockIdx.z, nsplit = gridDim.x;
    if (ld_flag(done + b)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c = lane & 31;
    // ---- quad geometry (uniform) ----
    // XOR steps (the whole schedule when ns is a power of two; the steps inside the groups of the grouped schedule): super-panels
    // {a, a^D, a^E, a^D^E}.  Group-pair steps of the grouped schedule (sc.super_order == 2, 0-based step D-1 >= G-1, G = 2^sc.gb): round r pairs group gA with
    // gB for the G offsets s (A_i <-> B_(i^s)); two consecutive offsets s, s' of a round close the quads {A_i, B_(i^s), B_(i^s'), A_(i^e)},
    // e = s ^ s' — the same four slots with the same roles: slots (0,1) and (3,2) rotate now, slots (0,2) and (3,1) meet next.
    int P0, P1, P2, P3, kcur0, kcur1, knxt0, knxt1;
    bool swapB, swapD;
    if (sc.super_order == 2 && D > (1 << sc.gb) - 1) {
        const int gb = sc.gb, G = 1 << gb;
        const int st = D - 1 - (G - 1), rnd = st >> gb, s0 = st & (G - 1), s1 = (E - 1 - (G - 1)) & (G - 1), e = s0 ^ s1;
        const int m = quad >> (gb - 1);
        if (m >= sc.gm) return;
        const int gA = sc.gpair[rnd][m][0], gB = sc.gpair[rnd][m][1];
        const int i = insert_zero_bit(quad & ((G >> 1) - 1), 31 - __clz(e));
        P0 = G * gA + i; P1 = G * gB + (i ^ s0); P2 = G * gB + (i ^ s1); P3 = G * gA + (i ^ e);
        swapB = true; swapD = true;
        kcur0 = G * m + i; kcur1 = G * m + (i ^ e);
        knxt0 = kcur0; knxt1 = kcur1;
    } else {
        const int h1 = 31 - __clz(D);
        const int e1 = ((E >> h1) & 1) ? (E ^ D) : E;
        const int h2 = 31 - __clz(e1);
        const int a0 = insert_zero_bit(insert_zero_bit(quad, min(h1, h2)), max(h1, h2));
        P0 = a0; P1 = a0 ^ D; P2 = a0 ^ E; P3 = a0 ^ D ^ E;   // super-panels in tile slots 0..3
        // current pairs (step D): A = slots (0,1), B = slots (2,3); Q order is (lower, upper) = (bit h1 clear, set)
        swapB = (P2 >> h1) & 1;
        const int hE = 31 - __clz(E);
        swapD = (P1 >> hE) & 1;
        kcur0 = remove_bit(P0, h1); kcur1 = remove_bit(swapB ? P3 : P2, h1);
        knxt0 = remove_bit(P0, hE); knxt1 = remove_bit(swapD ? P3 : P1, hE);
    }
    auto Pof = [&](int slot) { return slot == 0 ? P0 : (slot == 1 ? P1 : (slot == 2 ? P2 : P3)); };  // no indexed arrays: they go to scratch
    const int curS1 = swapB ? 3 : 2, curT1 = swapB ? 2 : 3;  // pair A: S = slot 0, T = slot 1
    const bool act0 = P0 < ns && P1 < ns && pair_active(subact, (int64_t)b * npairs + kcur0);
    const bool act1 = P2 < ns && P3 < ns && pair_active(subact, (int64_t)b * npairs + kcur1);
    // next pairs (step E): C = slots (0,2), D' = slots (1,3)
    const int nxtS1 = swapD ? 3 : 1, nxtT1 = swapD ? 1 : 3;  // pair C: S = slot 0, T = slot 2
    const bool have0 = P0 < ns && P2 < ns, have1 = P1 < ns && P3 < ns;
    if (chunk == 0 && tid == 0) {
        const int n = (act0 ? 1 : 0) + (act1 ? 1 : 0);
        if (n) atomicAdd(&nupd[b], n);  // instrumentation: super-pairs updated in this sweep
    }

    float* __restrict__ Xb = X + (int64_t)b * batch_stride;
    const int mypr = w >> 2, ob = w & 3;                      // this wave's pair and output block in Q order
    const int mS = mypr ? curS1 : 0, mT = mypr ? curT1 : 1;   // tile slots of this wave's pair in Q order
    const int oslot = ob < 2 ? mS : mT;                       // tile slot and panel (0..7) it owns
    const int otp = 2 * oslot + (ob & 1);
    const bool mine = mypr ? act1 : act0;
    float* __restrict__ Pw = Xb + (int64_t)(2 * min(Pof(oslot), ns - 1) + (ob & 1)) * panel_stride;

    // ---- column scales of this wave's pair ----
    // din[k], k in Q order (S0 S1 T0 T1); a member beyond ns holds nothing: its columns get the floor of the pair
    const float* __restrict__ dinp = Din + ((int64_t)b * npairs + (mypr ? kcur1 : kcur0)) * 128;
    const bool presS = Pof(mS) < ns, presT = Pof(mT) < ns;
    auto ein_raw = [&](int k) { return ((k < 64) ? presS : presT) ? sg_half_exp(dinp[k]) : -200; };
    int emax;
    {
        int e0 = max(ein_raw(lane), ein_raw(64 + lane));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e0 = max(e0, __shfl_xor(e0, o, 64));
        emax = e0 < -150 ? SG_TARGET : e0;   // nothing usable in the whole pair (all-zero columns): any scale does
    }
    auto ein_of = [&](int k) { return max(ein_raw(k), emax - SG_SPAN) - SG_TARGET; };

    // B operand of k-step s: lane (j = c, group h) holds Qs[16 s + 8 h + e][32 ob + c], e = 0..7, Qs = 2^ein Q 2^-eout, in two fp16 parts
    u32x4 q1[8], q2[8];
    float oscale;   // 2^eout of this lane's output column
    int eout;
    {
        const float* __restrict__ Qp = Qfin + ((int64_t)b * npairs + (mypr ? kcur1 : kcur0)) * (SP * SP);
        float v[8][8];
        float tmax = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s + 8 * h + e;
                const float qv = mine ? Qp[k * SP + 32 * ob + c] : ((k == 32 * ob + c) ? 1.0f : 0.0f);
                v[s][e] = ldexpf(qv, ein_of(k));
                tmax = fmaxf(tmax, fabsf(v[s][e]));
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));   // the other half of the k range sits in lane c of the other half-wave
        {
            const int bits = __float_as_int(tmax), ex = (bits >> 23) & 255;
            eout = (tmax > 0.0f && ex != 0 && ex != 255) ? ex - 126 : 0;   // tmax = f 2^eout, f in [0.5, 1); NaN / Inf / 0: leave as is (NaN propagates)
        }
        oscale = ldexpf(1.0f, eout);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                unsigned x, y;
                split2(ldexpf(v[s][2 * e2], -eout), ldexpf(v[s][2 * e2 + 1], -eout), x, y);
                q1[s][e2] = x; q2[s][e2] = y;
            }
        }
    }

    u32x4* aimg = (u32x4*)sg_smem;
    u32x4* opnd0 = (u32x4*)(sg_smem + SUPGRAM_AIMG_WORDS);
    constexpr int OPND_VECS = SUPGRAM_OPND_WORDS / 4;  // 16-byte operands per buffer
    int* egtab = (int*)(sg_smem + SUPGRAM_MAIN_FLOATS);   // eout of the 8 x 32 output columns (the scale of the Gram operands)
    if (!h) egtab[otp * 32 + c] = eout;

    // ---- this wave's three Gram half-tiles: unit u = w + 8 j -> tile u >> 1 (0..5 pair C, 6..11 pair D'), k-step u & 1 ----
    // sgram6 tile order [0,2] [0,3] [1,2] [1,3] [0,1] [2,3] over the pair's panels (0,1 = lower super-panel, 2,3 = upper)
    auto tile_panels = [&](int tt, int& pa, int& pb, bool& on) {
        const int np = tt >= 6 ? 1 : 0, t6 = tt - 6 * np;
        const int xa = t6 < 2 ? 0 : (t6 < 4 ? 1 : (t6 == 4 ? 0 : 2));
        const int xb = t6 == 0 ? 2 : (t6 == 1 ? 3 : (t6 == 2 ? 2 : (t6 == 3 ? 3 : (t6 == 4 ? 1 : 3))));
        const int sS = np ? nxtS1 : 0, sT = np ? nxtT1 : 2;
        pa = 2 * (xa < 2 ? sS : sT) + (xa & 1);
        pb = 2 * (xb < 2 ? sS : sT) + (xb & 1);
        on = np ? have1 : have0;
    };
    int ga0, gb0, ga1, gb1, ga2, gb2;
    bool gon0, gon1, gon2;
    tile_panels((w + 0) >> 1, ga0, gb0, gon0);
    tile_panels((w + 8) >> 1, ga1, gb1, gon1);
    tile_panels((w + 16) >> 1, ga2, gb2, gon2);
    const int kh = w & 1;
    f32x16 g0 = {0}, g1 = {0}, g2 = {0};

    const int r_begin = chunk * rows_per_wg;
    const int r_end = min(r_begin + rows_per_wg, R);
    const int ntiles = (r_end - r_begin) / 32;   // R and rows_per_wg are multiples of 32
    if (ntiles > 0) {
        // a thread fetches two 8-column pieces (32 B) of ITS PAIR's 32 x 128 tile: piece q = (tid & 255) + 256 jj -> panel q >> 7 of the pair in Q
        // order (0, 1: lower super-panel S, 2, 3: upper super-panel T), row (q & 127) >> 2, columns 8 (q & 3) .. +7; as an A operand that is
        // k-step / lane group (k >> 4, (k >> 3) & 1), k = 32 (q >> 7) + 8 (q & 3) its column in Q order
        f32x4 preA[2][2];   // one tile in flight per pair and thread (two in flight — 64 KB of loads per CU — measured slower: 1008 vs 895 us per launch)
        float mul[2][8];   // 2^-ein of the piece's eight columns
        int dst[2];
        const float* src[2];   // per-lane pointers of the two pieces in tile 0
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int q = (tid & 255) + 256 * jj, lp = q >> 7, row = (q & 127) >> 2;
            const int k = 32 * lp + 8 * (q & 3);
            dst[jj] = ((mypr * 8 + (k >> 4)) * 2) * SG_BLK + ((k >> 3) & 1) * SG_HB + row;
            const int sp = Pof(jj ? mT : mS);   // jj = 0: pieces of S (lp = 0, 1), jj = 1: pieces of T (lp = 2, 3)
            src[jj] = sp < ns ? Xb + (int64_t)(2 * sp + (lp & 1)) * panel_stride + (int64_t)r_begin * PB + (q & 127) * 8 : nullptr;
#pragma unroll
            for (int e = 0; e < 8; ++e) mul[jj][e] = ldexpf(1.0f, -ein_of(k + e));
        }
        auto fetch = [&](int t, f32x4 (&pre)[2][2]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (src[jj]) {
                    const float* s_ = src[jj] + (int64_t)t * (32 * PB);
                    pre[jj][0] = *(const f32x4*)(s_);
                    pre[jj][1] = *(const f32x4*)(s_ + 4);
                } else {
                    pre[jj][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    pre[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        auto stash = [&](const f32x4 (&pre)[2][2]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                u32x4 p1, p2;
                unsigned x, y;
                split2s(pre[jj][0][0], pre[jj][0][1], mul[jj][0], mul[jj][1], x, y); p1[0] = x; p2[0] = y;
                split2s(pre[jj][0][2], pre[jj][0][3], mul[jj][2], mul[jj][3], x, y); p1[1] = x; p2[1] = y;
                split2s(pre[jj][1][0], pre[jj][1][1], mul[jj][4], mul[jj][5], x, y); p1[2] = x; p2[2] = y;
                split2s(pre[jj][1][2], pre[jj][1][3], mul[jj][6], mul[jj][7], x, y); p1[3] = x; p2[3] = y;
                u32x4* o = aimg + dst[jj];
                o[0] = p1; o[SG_BLK] = p2;
            }
        };
        // The matrix instructions of a compute segment — 9 for the Gram half-tiles of an earlier tile, 24 for the update — read all their A
        // operands (and the Gram B operands) from LDS.  Left to the compiler every ds_read sits right in front of its consumer (s_waitcnt
        // lgkmcnt(0) before each MFMA group) and the LDS latency is exposed; here the reads of stage i+1 are issued before the MFMAs of stage i,
        // pinned with sched_barrier.  Consecutive matrix instructions never share an accumulator (a dependent pair costs an issue bubble):
        // the Gram units take turns; the update keeps ONE accumulator (its compute segment is the shorter one, see the panel stores below).
        struct Opnd4 { h16x8 a1, a2, b1, b2; };
        struct Opnd2 { h16x8 a1, a2; };
        auto ldG = [&](const u32x4* op, int pa, int pb) {
            const u32x4* oa = op + ((pa * 2 + kh) * 2) * 64 + lane;
            const u32x4* ob_ = op + ((pb * 2 + kh) * 2) * 64 + lane;
            Opnd4 r;
            r.a1 = __builtin_bit_cast(h16x8, oa[0]); r.a2 = __builtin_bit_cast(h16x8, oa[64]);
            r.b1 = __builtin_bit_cast(h16x8, ob_[0]); r.b2 = __builtin_bit_cast(h16x8, ob_[64]);
            return r;
        };
        auto gram3 = [&](const Opnd4& x, const Opnd4& y, const Opnd4& z) {   // small terms first, units interleaved
            if (gon0) g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.a2, x.b1, g0, 0, 0, 0);
            if (gon1) g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y.a2, y.b1, g1, 0, 0, 0);
            if (gon2) g2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(z.a2, z.b1, g2, 0, 0, 0);
            if (gon0) g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.a1, x.b2, g0, 0, 0, 0);
            if (gon1) g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y.a1, y.b2, g1, 0, 0, 0);
            if (gon2) g2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(z.a1, z.b2, g2, 0, 0, 0);
            if (gon0) g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.a1, x.b1, g0, 0, 0, 0);
            if (gon1) g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y.a1, y.b1, g1, 0, 0, 0);
            if (gon2) g2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(z.a1, z.b1, g2, 0, 0, 0);
        };
        auto gram_of = [&](int t) {  // the three units of tile t, nothing to overlap with (tail)
            if (r_begin + 32 * t >= m_pad) return;
            const u32x4* op = opnd0 + (t & 1) * OPND_VECS;
            const Opnd4 x = ldG(op, ga0, gb0), y = ldG(op, ga1, gb1), z = ldG(op, ga2, gb2);
            gram3(x, y, z);
        };
        const u32x4* img = aimg + (mypr * 8 * 2) * SG_BLK + h * SG_HB + c;
        auto ldA = [&](int s) {
            Opnd2 r;
            r.a1 = __builtin_bit_cast(h16x8, img[(2 * s + 0) * SG_BLK]);
            r.a2 = __builtin_bit_cast(h16x8, img[(2 * s + 1) * SG_BLK]);
            return r;
        };
        const int glag = mypr ? 1 : 2;   // pair A accumulates the Gram units of tile t - 2 next to update t, pair B those of tile t - 1

        // ---- prologue: tile 0 staged, tile 1 in flight ----
        fetch(0, preA);
        stash(preA);
        if (ntiles > 1) fetch(1, preA);
        __syncthreads();
        if (mypr) __syncthreads();   // pair B runs one segment behind pair A
#ifdef ASVD_SG_TIMING
        unsigned long long sg_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sg_last = __builtin_amdgcn_s_memtime();
#endif
        // one tile: compute segment, barrier, memory segment, barrier.  `pre` holds tile t + 1
        auto tile_step = [&](const int t, f32x4 (&pre)[2][2]) __attribute__((always_inline)) {
            // ================= compute segment =================
            SG_TS(0);
            const int tg = t - glag;
            const bool pending = tg >= 0 && r_begin + 32 * tg < m_pad;   // rows of the matrix proper only, not accumulated V rows
            Opnd2 ua;
            if (pending) {
                const u32x4* op = opnd0 + (tg & 1) * OPND_VECS;
                const Opnd4 x = ldG(op, ga0, gb0), y = ldG(op, ga1, gb1), z = ldG(op, ga2, gb2);
                ua = ldA(0);
                __builtin_amdgcn_sched_barrier(0);
                if (!SG_ABL(4)) gram3(x, y, z);
            } else {
                ua = ldA(0);
            }
            SG_TS(1);
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                Opnd2 un = ua;
                if (s + 1 < 8) un = ldA(s + 1);
                __builtin_amdgcn_sched_barrier(0);
                const h16x8 B1 = __builtin_bit_cast(h16x8, q1[s]), B2 = __builtin_bit_cast(h16x8, q2[s]);
                if (!SG_ABL(4)) {   // one accumulator, small terms first (two alternating accumulators + their sum: 892.5 vs 894.3 us per launch, nothing)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.a2, B1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.a1, B2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.a1, B1, acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                ua = un;
            }
            SG_TS(2);
            __syncthreads();   // own pair: done reading the incoming image; other pair: its memory segment ends
            SG_TS(3);
            // ================= memory segment =================
            // order matters: vmcnt counts loads AND stores, and the panel stores are conditional (the compiler must assume none were issued), so a
            // stash BEHIND this tile's stores would wait for them.  Stash first: it waits only for what the previous memory segment issued.
            if (t + 1 < ntiles && !SG_ABL(8)) stash(pre);     // tile t + 1 (in registers since the previous memory segment) -> incoming image
            SG_TS(4);
            if (t + 2 < ntiles && !SG_ABL(2)) fetch(t + 2, pre);   // into the registers the stash has just emptied
            SG_TS(5);
            SG_TS(6);
            if (r_begin + 32 * t < m_pad && !SG_ABL(8)) {
                u32x4* ow = opnd0 + (t & 1) * OPND_VECS;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    u32x4 p1, p2;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        unsigned x, y;
                        split2(acc[8 * k2 + 2 * e2], acc[8 * k2 + 2 * e2 + 1], x, y);
                        p1[e2] = x; p2[e2] = y;
                    }
                    u32x4* o = ow + ((otp * 2 + k2) * 2) * 64 + lane;
                    o[0] = p1; o[64] = p2;
                }
            }
            if (mine && !SG_ABL(1)) {
                // Panel stores, LAST in the segment and with the address formed per tile: sixteen 4-byte stores per lane (two full 128-byte rows per
                // instruction).  This kernel waits for its vector-memory pipe (no stores: 769 instead of 1011 us per launch), and what was measured there
                // does not follow a model: the same sixteen stores right behind the fetch cost 995 us per launch when their per-lane address is kept in a
                // register pair across the loop, 893 when it is recomputed in front of them (four more VALU instructions), 884 when they are issued after
                // the operand split — reproduced binary by binary on one box (profiles/r4_supgram_variants.txt).  Measured and dropped: the tile through
                // 4 KB of wave-private LDS and out as four 16-byte row stores per lane (1047 us); 4 x 4 transposes inside the lane quads (DPP quad_perm, 64
                // VALU per tile) and four 16-byte stores per lane (889 vs 884: the store COUNT is not what it waits for); two tiles of loads in flight — left to the
                // compiler 1008 us (it waits for ALL loads, vmcnt(0)), with hand-issued loads and exact s_waitcnt vmcnt(36) counts 948 vs 957 on the same box:
                // more bytes in flight buy nothing (profiles/r4_supgram_variants.txt); a panel stride that is not a power of two (+24 rows): 937 vs 957.
                float* __restrict__ po = Pw + (int64_t)(r_begin + 32 * t + 4 * h) * PB + c;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) po[((reg & 3) + 8 * (reg >> 2)) * PB] = acc[reg] * oscale;
            }
            SG_TS(7);
            __syncthreads();
            SG_TS(8);
        };
        for (int t = 0; t < ntiles; ++t) tile_step(t, preA);
#ifdef ASVD_SG_TIMING
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && ob == 0 && lane == 0) {
            for (int i = 0; i < 9; ++i) g_sg_ts[mypr][i] = sg_acc[i];
            g_sg_ts[mypr][9] = (unsigned long long)ntiles;
        }
#endif
        if (!mypr) __syncthreads();   // pair A waits for pair B's last memory segment
        // ---- tail: the Gram units not yet accumulated (pair A: the last two tiles, pair B: the last one) ----
        if (ntiles - glag >= 0) gram_of(ntiles - glag);
        if (glag == 2) gram_of(ntiles - 1);
    }

    // ---- the two k-steps of a tile sit in waves 2t and 2t+1 (j = 0), 2t-8 .. (j = 1), ...: sum through LDS, rescale, coalesced store ----
    __syncthreads();
    float* red = sg_smem;  // [unit u = 0..23][16 x 64]
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        red[(w + 0) * 1024 + reg * 64 + lane] = g0[reg];
        red[(w + 8) * 1024 + reg * 64 + lane] = g1[reg];
        red[(w + 16) * 1024 + reg * 64 + lane] = g2[reg];
    }
    __syncthreads();
    for (int o = tid; o < 12 * 1024; o += 512) {
        const int tt = o >> 10, e = o & 1023, np = tt >= 6 ? 1 : 0;
        if (!(np ? have1 : have0)) continue;
        int pa, pb;
        bool on;
        tile_panels(tt, pa, pb, on);
        const int reg = e >> 6, ln = e & 63;
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), j = ln & 31;
        const float v = ldexpf(red[(2 * tt) * 1024 + e] + red[(2 * tt + 1) * 1024 + e], egtab[pa * 32 + i] + egtab[pb * 32 + j]);
        Gx[((((int64_t)b * npairs + (np ? knxt1 : knxt0)) * nsplit + chunk) * 6 + (tt - 6 * np)) * 1024 + i * 32 + j] = v;
    }
}
