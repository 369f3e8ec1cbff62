// svd_jacobi.hip — K4: batched economy SVD in fp32 by one-sided block Jacobi, written for gfx950.
//
// Replaces the factorisation at modules/svd_linear.py:65 (oracle: torch.linalg.svd, BASELINE.json) and the
// values-only torch.svd at sensitivity.py:101.  See DESIGN.md §"K4" for the algorithm and roofline.
//
// Data layout in HBM (per problem): the oriented matrix (rows >= cols) is stored as nb = ncols_pad/32
// column panels.  Panel I is a dense [R][32] fp32 array (row stride 128 B = one cache line), so a wave
// reads two consecutive rows of a panel with ONE fully coalesced 256-B load, and a panel pair is two
// contiguous streams.  Rows [0, m_pad) hold A (times the column scale), rows [m_pad, m_pad + n_pad) hold
// the accumulated right factor V (identity at start) so that one update kernel rotates both.
//
// Per step of the pair schedule (XOR ordering, see rr_pair; disjoint panel pairs) three launches:
//   gram_kernel    G = [A_I A_J]^T [A_I A_J]  (64x64; blocks II, IJ, JJ) — v_mfma_f32_32x32x2_f32, K = rows
//   evd_kernel     two-sided Jacobi on G in LDS (fp32), eigenvalues sorted descending -> Q (64x64)
//   update_kernel  [X_I X_J] <- [X_I X_J] * Q  over all R rows — v_mfma_f32_32x32x2_f32, K = 64
#include "common.h"
#include "jacobi_shared.h"
#include <vector>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>

namespace {
using namespace asvdk;


// --------------------------------------------------------------------------------------------------
// pack: oriented, scaled, fp32 copy of the input into panel layout.  X must be zero-filled before.
//   transposed == 0:  X[blk][r][c] = float(src[r][blk*32+c]) * float(s[blk*32+c])      r < rows, col < cols
//   transposed == 1:  X[blk][r][c] = float(src[blk*32+c][r]) * float(s[r])             (oriented = src^T)
// The product is the reference's `w.float() * s.view(1,-1)` (fp32 * upcast(s)), svd_linear.py:47,60.
template <int DT, int ST>
__global__ __launch_bounds__(256) void pack_kernel(const void* __restrict__ src, int64_t ld, const void* __restrict__ s,
                                                   int has_scale, int transposed, int rows, int cols, int R,
                                                   float* __restrict__ X) {
    // block: 32 oriented rows x 256 oriented columns (8 panels)
    const int tx = threadIdx.x;
    const int r0 = blockIdx.x * 32;
    const int c0 = blockIdx.y * 256;
    if (!transposed) {
        const int col = c0 + tx;
        if (col >= cols) return;
        const float sc = has_scale ? elem<ST>::ld(s, col) : 1.0f;
        float* dst = X + ((int64_t)(col >> 5) * R) * PB + (col & 31);
        for (int i = 0; i < 32; ++i) {
            const int r = r0 + i;
            if (r >= rows) break;
            dst[(int64_t)r * PB] = elem<DT>::ld(src, (int64_t)r * ld + col) * sc;
        }
    } else {
        // tile transpose through LDS: read src[c][r] coalesced along r, write X[.][r][c] coalesced along c
        __shared__ float tile[32][33];
        const int lx = tx & 31, ly = tx >> 5;  // 32 x 8
        for (int p = 0; p < 8; ++p) {          // 8 panels of 32 oriented columns
            const int cb = c0 + p * 32;
            if (cb >= cols) break;             // uniform per block
            for (int j = ly; j < 32; j += 8) {
                const int c = cb + j, r = r0 + lx;
                float v = 0.0f;
                if (c < cols && r < rows) {
                    const float sc = has_scale ? elem<ST>::ld(s, r) : 1.0f;
                    v = elem<DT>::ld(src, (int64_t)c * ld + r) * sc;
                }
                tile[j][lx] = v;
            }
            __syncthreads();
            float* dst = X + ((int64_t)(cb >> 5) * R) * PB;
            for (int i = ly; i < 32; i += 8) {
                const int r = r0 + i;
                if (r < rows && cb + lx < cols) dst[(int64_t)r * PB + lx] = tile[lx][i];
            }
            __syncthreads();
        }
    }
}

// V part: identity on the real columns
__global__ void vinit_kernel(float* __restrict__ X, int cols, int R, int m_pad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < cols) X[((int64_t)(j >> 5) * R + m_pad + j) * PB + (j & 31)] = 1.0f;
}

// --------------------------------------------------------------------------------------------------
// gram: per (row split, pair, problem) partial 64x64 Gram matrix, three 32x32 blocks II, IJ, JJ.
// MFMA 32x32x2 f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; with A = panel^T the operand of a wave for
// rows (r, r+1) is simply panel[r*32 + l].  Panels are streamed HBM -> registers (16-B loads, the next 32-row chunk prefetched while
// the current one is in the matrix pipe) -> a wave-private 8-KiB LDS image of the HBM layout (32 rows of both panels), then read
// back as conflict-free ds_read_b32 (one per MFMA operand).
constexpr int GCH = 32;  // rows per staged chunk


__global__ __launch_bounds__(256) void gram_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride,
                                                   int nb, int step, int m_pad, int rows_per_split,
                                                   float* __restrict__ Gpart, const int* __restrict__ done,
                                                   const int* __restrict__ plist, int list_stride) {
    const int split = blockIdx.x, pair = blockIdx.y, b = blockIdx.z;
    const int nsplit = gridDim.x, npairs = gridDim.y;
    ASVD_KERNEL_ACQUIRE(sc);
    if (ld_flag(done + b)) return;
    int I, J;
    if (!get_pair(sc, plist, list_stride, b, nb, step, pair, I, J)) return;
    const float* __restrict__ XI = X + (int64_t)b * batch_stride + (int64_t)I * panel_stride;
    const float* __restrict__ XJ = X + (int64_t)b * batch_stride + (int64_t)J * panel_stride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r_begin = split * rows_per_split;
    const int r_end = min(r_begin + rows_per_split, m_pad);
    const int nchunks = (r_end - r_begin) / GCH;  // m_pad and rows_per_split are multiples of 32

    __shared__ __attribute__((aligned(16))) float stage[4][2 * GCH * PB];  // per wave: panel I chunk, panel J chunk
    float* sI = stage[w];
    float* sJ = stage[w] + GCH * PB;

    f32x16 aii = {0}, aij = {0}, ajj = {0};
    {
        // register prefetch: the next chunk's 8 KB are in flight (32 VGPRs) while this chunk is in the matrix pipe, which
        // doubles the bytes a wave keeps outstanding compared with staging by LDS-DMA and waiting (measured 196 -> 177 us per launch)
        f32x4 pI[GCH / 8], pJ[GCH / 8];
        auto fetch = [&](int ch) {
            const int64_t r0 = r_begin + (int64_t)ch * GCH;
#pragma unroll
            for (int it = 0; it < GCH / 8; ++it) {
                pI[it] = *(const f32x4*)(XI + (r0 + it * 8) * PB + lane * 4);
                pJ[it] = *(const f32x4*)(XJ + (r0 + it * 8) * PB + lane * 4);
            }
        };
        if (w < nchunks) fetch(w);
        for (int ch = w; ch < nchunks; ch += 4) {
#pragma unroll
            for (int it = 0; it < GCH / 8; ++it) {
                *(f32x4*)(sI + it * 256 + lane * 4) = pI[it];
                *(f32x4*)(sJ + it * 256 + lane * 4) = pJ[it];
            }
            if (ch + 4 < nchunks) fetch(ch + 4);
#pragma unroll
            for (int u = 0; u < GCH / 2; ++u) {
                const float a = sI[u * 64 + lane];
                const float c = sJ[u * 64 + lane];
                aii = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, aii, 0, 0, 0);
                aij = __builtin_amdgcn_mfma_f32_32x32x2f32(a, c, aij, 0, 0, 0);
                ajj = __builtin_amdgcn_mfma_f32_32x32x2f32(c, c, ajj, 0, 0, 0);
            }
        }
    }

    // cross-wave reduction in a fixed order ((w0 + w2) + (w1 + w3)), reusing the staging LDS (2 x 12 KiB), then wave 0
    // stores the natural [t][i][j] layout straight from its accumulators (2 rows x 128 B per store instruction).
    __syncthreads();
    float* red = &stage[0][0];
    if (w >= 2) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            red[((w - 2) * 48 + 0 + reg) * 64 + lane] = aii[reg];
            red[((w - 2) * 48 + 16 + reg) * 64 + lane] = aij[reg];
            red[((w - 2) * 48 + 32 + reg) * 64 + lane] = ajj[reg];
        }
    }
    __syncthreads();
    if (w < 2) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            aii[reg] += red[(w * 48 + 0 + reg) * 64 + lane];
            aij[reg] += red[(w * 48 + 16 + reg) * 64 + lane];
            ajj[reg] += red[(w * 48 + 32 + reg) * 64 + lane];
        }
    }
    __syncthreads();
    if (w == 1) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            red[(0 + reg) * 64 + lane] = aii[reg];
            red[(16 + reg) * 64 + lane] = aij[reg];
            red[(32 + reg) * 64 + lane] = ajj[reg];
        }
    }
    __syncthreads();
    if (w == 0) {
        float* out = Gpart + (((int64_t)b * npairs + pair) * nsplit + split) * 3072;
        const int h = lane >> 5, c = lane & 31;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            out[0 * 1024 + i * 32 + c] = aii[reg] + red[(0 + reg) * 64 + lane];
            out[1 * 1024 + i * 32 + c] = aij[reg] + red[(16 + reg) * 64 + lane];
            out[2 * 1024 + i * 32 + c] = ajj[reg] + red[(32 + reg) * 64 + lane];
        }
    }
    ASVD_KERNEL_RELEASE(sc);
}

// --------------------------------------------------------------------------------------------------
// evd: one 256-thread workgroup per pair.  Two-sided Jacobi on the 64x64 Gram matrix in LDS with the parallel
// round-robin ordering (63 steps per sweep, 32 disjoint rotations per step).  Thread (ty, tx) of a 16x16 grid owns the
// 2x2 blocks {row pairs 2ty, 2ty+1} x {column pairs 2tx, 2tx+1} and rows 4ty..4ty+3 of those column pairs of Q.
// Every thread recomputes the four rotations it needs from the OLD matrix (G is ping-ponged between two LDS images), so
// a step costs ONE barrier and no serialized "compute rotations" phase.  Pair tables are precomputed in LDS.

// LDS image: element (row r, position c) lives at r*64 + pcol(c), pcol(c) = (c&1)*32 + (c>>1)  ("plane-major" columns:
// even positions in banks 0..31 of a row, odd positions in the next 32).  With ONE column pair per lane (32 lanes of a
// half-wave = 32 pairs) every ds_read_b32/ds_write_b32 of the sweep is bank-conflict free, for both pairings below.
__device__ __forceinline__ int pcol(int c) { return ((c & 1) << 5) | (c >> 1); }

// Odd-even transposition ordering with swap (Luk-Park): phase A pairs positions (2k, 2k+1), phase B pairs (2k+1, 2k+2)
// (positions 63 and 0 idle); after each rotation the two columns/rows EXCHANGE places, so in 64 phases every pair of the 64
// columns has met exactly once (one sweep) and nothing ever moves in LDS except by the rotate-and-swap itself: all updates
// are in place, each 2x2 block is owned by exactly one thread per phase, one barrier per phase.
// Thread (g, tx): column pair tx, row pairs 4g..4g+3 (4 blocks of G), rows 8g..8g+7 of Q's column pair tx.
// Each lane computes the rotation of ITS column pair from a small side array (diagonal + pivot off-diagonals, ping-ponged);
// the four row-pair rotations are the ones lanes 4g..4g+3 of the same half-wave just computed -> fetched with ds_bpermute.
// Two-level sweeps (twolevel.h) run the same solver in two more modes; one workgroup per sub-pair of a super-pair (S, T) whose four
// 32-blocks are numbered S0, S1, T0, T1 = 0..3:
//   MODE 1 (inner step 0, sub-pairs (0,2) and (1,3)): the 64x64 matrix is assembled from the carried 32x32 diagonal blocks of the two
//           panels (v3.Gd32) and the summed cross tile of sgram6; outputs Q0 (sorted, normalised; identity when nothing rotates), the
//           two transformed diagonal blocks (v3.D0) and the activity flag;
//   MODE 2 (inner step 1, sub-pairs (0,3) and (1,2)): the diagonal blocks come from D0, the cross block is the transformed tile
//           Q0_a[:, :32]^T G[{0,2},{1,3}] Q0_b[:, 32:] (or its mirror), computed here with fp32 MFMA from sgram6's tiles; outputs the new
//           carried diagonal blocks of both panels (Gd32) and this sub-pair's 128x64 column block of Qfin = Q^(0) Q^(1), the matrix
//           supdate applies.  No 128x128 matrix is ever materialised and nothing else runs between the Gram pass and the update.
//   MODE 0 is the single-level solve (Gram partials of gram_kernel); with v3.Gd32 set (internal step d = 1 of a two-level sweep) it
//           also stores the two transformed diagonal blocks as the fresh carried blocks of its panels.
// A solve that does not rotate (all couplings below tol) leaves everything in place: identity Q, no sort.

// transformed diagonal 32x32 blocks of the (sorted, rescaled) matrix left in LDS: block h = sorted positions 32h..32h+31
__device__ __forceinline__ void store_diag_blocks(const float* G, const int* rnk, const float* cscale, float* d0, float* d1, int tid) {
    for (int e = tid; e < PW * PW; e += 256) {
        const int r = e >> 6, c = e & 63;
        const int rr = rnk[r], rc = rnk[c];
        if ((rr >> 5) == (rc >> 5)) {
            float* dst = (rr >> 5) ? d1 : d0;
            dst[(rr & 31) * 32 + (rc & 31)] = G[r * PW + pcol(c)] * cscale[r] * cscale[c];
        }
    }
}

// block coordinates of a kernel body: the bodies below run either as their own launch or as one half of a merged launch (dual
// kernels further down), so they take their grid position as data instead of reading blockIdx / gridDim
struct BlockCtx { int bx, by, bz, gx, gy, gz; };

constexpr int EVD_SMEM_FLOATS(int keepg) { return (keepg ? 2 : 1) * PW * PW + 2 * PW + 64 + 8 + PW + PW; }

template <int MODE, int KEEPG>
__device__ __forceinline__ void evd_body(const Sched& sc, const BlockCtx& ctx, float* __restrict__ smem, const float* __restrict__ Gpart, int nsplit,
                                         float* __restrict__ Qbuf, int* __restrict__ active, unsigned* __restrict__ maxoff_bits,
                                         int* __restrict__ nrot, const int* __restrict__ done, float tol, int inner_sweeps, int nb, int step,
                                         int kb, int* __restrict__ hist, const int* __restrict__ plist, int list_stride, const EvdV3& v3) {
    static_assert(MODE == 0 || KEEPG == 1, "the two-level modes read G after the solve");
    // LDS carve-up (EVD_SMEM_FLOATS): G, [Qs], sdiag[2][64], sb[2][32], redmax[4], redmax_t[4], cscale[64], rnk[64]
    float* G = smem;
    float* Qs = smem + PW * PW;
    float* Q = KEEPG ? Qs : G;
    float* small = smem + (KEEPG ? 2 : 1) * PW * PW;
    float (*sdiag)[PW] = (float (*)[PW])small;
    float (*sb)[32] = (float (*)[32])(small + 2 * PW);
    float* redmax = small + 2 * PW + 64;
    float* redmax_t = redmax + 4;
    float* cscale = redmax + 8;
    int* rnk = (int*)(cscale + PW);

    const int pair = MODE ? (ctx.bx >> 1) : ctx.bx, b = ctx.by, npairs = MODE ? (ctx.gx >> 1) : ctx.gx;
    ASVD_KERNEL_ACQUIRE(sc);
    if (sc.fence & 4) {  // experiment (ASVD_FENCE=4, tools/repro_two_streams.py): start from a zeroed LDS image
        for (int e = threadIdx.x; e < EVD_SMEM_FLOATS(KEEPG); e += 256) smem[e] = 0.0f;
        __syncthreads();
    }
    if (sc.dbg_fill) {  // experiment (ASVD_EVD_LDSFILL=mask): NaN into the regions G | Qs | sdiag | sb | redmax | cscale | rnk (bits 0..6)
        const int base = (KEEPG ? 2 : 1) * PW * PW;
        const int lo[7] = {0, PW * PW, base, base + 128, base + 192, base + 200, base + 264};
        const int hi[7] = {PW * PW, KEEPG ? 2 * PW * PW : PW * PW, base + 128, base + 192, base + 200, base + 264, base + 328};
        for (int r = 0; r < 7; ++r)
            if ((sc.dbg_fill >> r) & 1)
                for (int e = lo[r] + (int)threadIdx.x; e < hi[r]; e += 256) smem[e] = __builtin_nanf("");
        __syncthreads();
    }
    if (ld_flag(done + b)) return;
    // the eigen-solve is a dependent chain of short VALU/LDS phases on every group's critical path: let its waves win the issue
    // arbitration against the matrix-pipe-bound gram/update waves of the other stream groups that share the SIMD
    if (!(sc.fence & 8)) __builtin_amdgcn_s_setprio(3);   // (ASVD_FENCE=8: experiment without the priority)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sp = MODE ? (ctx.bx & 1) : 0;
    const int64_t slot = (int64_t)b * npairs + pair;
    int I, J;        // the two 32-column panels of this solve
    int* act_flag;   // where this solve reports whether it rotated
    float* qo;       // its 64x64 Q (MODE 0, 1)
    int S = 0, T = 0;
    if constexpr (MODE == 0) {
        act_flag = active + b * npairs + pair;
        qo = Qbuf + slot * (PW * PW);
        if (!get_pair(sc, plist, list_stride, b, nb, step, pair, I, J)) {  // padding pair / empty slot: nothing to rotate
            if (tid == 0) *act_flag = 0;
            return;
        }
        const float* gp = Gpart + slot * nsplit * 3072;
        // sum the row-split partials in fixed order: the three stored 32x32 blocks (II, IJ, JJ) are read fully coalesced (12
        // independent elements per thread keep 12+ loads in flight per split); the JI block is the mirror of IJ, written to LDS twice
        float acc[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
#pragma unroll 2
        for (int s2 = 0; s2 < nsplit; ++s2) {
#pragma unroll
            for (int q = 0; q < 12; ++q) acc[q] += gp[(int64_t)s2 * 3072 + tid + 256 * q];
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int o = tid + 256 * q;
            const int t = o >> 10, ii = (o & 1023) >> 5, jj = o & 31;
            const int i = ii + (t == 2 ? 32 : 0), j = jj + (t == 0 ? 0 : 32);
            G[i * PW + pcol(j)] = acc[q];
            if (t == 1) G[j * PW + pcol(i)] = acc[q];
        }
    } else {
        act_flag = v3.subact + slot * 4 + (MODE - 1) * 2 + sp;
        qo = v3.Q0 + (slot * 2 + sp) * (PW * PW);
        super_pair(sc, v3.ns, step, pair, S, T);
        if (T >= v3.ns) {  // padding super-pair
            if (tid == 0) *act_flag = 0;
            return;
        }
        // blocks (a, b) of this solve: step 0: (0,2),(1,3); step 1: (0,3),(1,2)
        const int ba = sp, bb = (MODE == 1) ? 2 + sp : 3 - sp;
        I = 2 * S + ba;
        J = 2 * T + (bb - 2);
        const float* __restrict__ gx = v3.Gx6 + slot * v3.nsplit6 * (6 * 1024);
        const float *dA, *dB;  // the two diagonal blocks
        if constexpr (MODE == 1) {
            dA = v3.Gd32 + ((int64_t)b * v3.nbpan + I) * 1024;
            dB = v3.Gd32 + ((int64_t)b * v3.nbpan + J) * 1024;
            // cross block = summed tile [0,2] (tile 0) or [1,3] (tile 3)
            const int tile = sp ? 3 : 0;
            // partials outermost: the four loads of a partial are independent and several partials are in flight (the sum of an element
            // still runs over the partials in ascending order).  With the loop nest the other way round every element waited for its
            // partials one L2 round trip at a time: 49 -> 69 ms of solves per batch-1 SVD between 2 and 16 partials.
            float v4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
            for (int s2 = 0; s2 < v3.nsplit6; ++s2)
#pragma unroll
                for (int q = 0; q < 4; ++q) v4[q] += gx[(int64_t)s2 * 6144 + tile * 1024 + tid + 256 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q, i = e >> 5, j = e & 31;
                G[i * PW + pcol(32 + j)] = v4[q];
                G[(32 + j) * PW + pcol(i)] = v4[q];
            }
        } else {
            dA = v3.D0 + (slot * 4 + ba) * 1024;
            dB = v3.D0 + (slot * 4 + bb) * 1024;
            // cross block C = QA[:, :32]^T MM QB[:, 32:], MM = M = G[{0,2},{1,3}] (sp 0, QA = Q0_0, QB = Q0_1) or M^T (sp 1, swapped).
            // LDS: MMt (the transpose of MM, 64x64) in Qs; QAh (64x32) and QBh (64x32) in G; T = MM QBh goes over MMt.
            float* MMt = Qs;
            float* QAh = G;
            float* QBh = G + 2048;
            const float* __restrict__ qa = v3.Q0 + (slot * 2 + sp) * (PW * PW);
            const float* __restrict__ qb = v3.Q0 + (slot * 2 + (sp ^ 1)) * (PW * PW);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + 256 * q, k = e >> 5, i = e & 31;
                QAh[e] = qa[k * PW + i];
                QBh[e] = qb[k * PW + 32 + i];
            }
            // M blocks from the summed tiles: [0,1] = tile 4, [0,3] = tile 1, [2,1] = tile 2 ^T, [2,3] = tile 5
            for (int qg = 0; qg < 4; ++qg) {  // four elements at a time (register budget of 4 workgroups per CU), partials outermost as above
                int off4[4];
                float v4[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int e = tid + 256 * (4 * qg + q4), k = e >> 6, i = e & 63;  // MMt[k][i]
                    // sp 0: MMt[k][i] = M[i][k] (i: rows = blocks 0,2; k: columns = blocks 1,3);  sp 1: MMt[k][i] = M[k][i]
                    const int mr = sp ? k : i, mc = sp ? i : k;
                    const int rb = mr >> 5, cb = mc >> 5, ri = mr & 31, ci = mc & 31;
                    if (rb == 0 && cb == 0) off4[q4] = 4 * 1024 + ri * 32 + ci;
                    else if (rb == 0) off4[q4] = 1 * 1024 + ri * 32 + ci;
                    else if (cb == 0) off4[q4] = 2 * 1024 + ci * 32 + ri;  // [2,1] = [1,2]^T
                    else off4[q4] = 5 * 1024 + ri * 32 + ci;
                    v4[q4] = 0.0f;
                }
#pragma unroll 4
                for (int s2 = 0; s2 < v3.nsplit6; ++s2)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) v4[q4] += gx[(int64_t)s2 * 6144 + off4[q4]];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int e = tid + 256 * (4 * qg + q4);
                    MMt[(e >> 6) * PW + (e & 63)] = v4[q4];
                }
            }
            __syncthreads();
            const int hh = lane >> 5, cc = lane & 31;
            f32x16 acc = {0};
            if (wv < 2) {  // T[32 wv + i][j] = sum_k MM[32 wv + i][k] QBh[k][j]
#pragma unroll 8
                for (int k2 = 0; k2 < 32; ++k2) {
                    const int k = 2 * k2 + hh;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(MMt[k * PW + 32 * wv + cc], QBh[k * 32 + cc], acc, 0, 0, 0);
                }
            }
            __syncthreads();  // MMt fully consumed
            if (wv < 2) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    MMt[(32 * wv + i) * 32 + cc] = acc[reg];  // T, row-major 64 x 32
                }
            }
            __syncthreads();
            acc = (f32x16){0};
            if (wv == 0) {  // C[i][j] = sum_k QAh[k][i] T[k][j]
#pragma unroll 8
                for (int k2 = 0; k2 < 32; ++k2) {
                    const int k = 2 * k2 + hh;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(QAh[k * 32 + cc], MMt[k * 32 + cc], acc, 0, 0, 0);
                }
            }
            __syncthreads();  // QAh / QBh (the G region) are free now
            if (wv == 0) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    G[i * PW + pcol(32 + cc)] = acc[reg];
                    G[(32 + cc) * PW + pcol(i)] = acc[reg];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, i = e >> 5, j = e & 31;
            G[i * PW + pcol(j)] = dA[e];
            G[(32 + i) * PW + pcol(32 + j)] = dB[e];
        }
    }
    __syncthreads();
    if (tid < PW) sdiag[0][tid] = G[tid * PW + pcol(tid)];
    if (tid < 32) sb[0][tid] = G[(2 * tid) * PW + pcol(2 * tid + 1)];  // phase A pivots G[2k][2k+1]
    __syncthreads();

    // scaled off-diagonal measure: max |g_ij| / sqrt(g_ii g_jj), NaN propagating.
    // `loc` decides whether the pair rotates; `loct` (entries touching a LEADING panel, index < kb) is what termination
    // looks at: the caller asked for k leading triplets, pair sorting keeps the largest columns in the lowest panels, and the
    // leading columns only need to be orthogonal among themselves and to the tail's SPAN - tail-internal angles (the JJ block
    // of a leading/tail pair, or a tail/tail pair) keep being rotated but no longer hold up termination.
    const bool topI = I < kb, topJ = J < kb;
    float loc = 0.0f, loct = 0.0f;
    for (int e = tid; e < PW * PW; e += 256) {
        const int i = e >> 6, j = e & 63;
        if (i != j) {
            const float dd = sdiag[0][i] * sdiag[0][j];
            const float g = G[i * PW + pcol(j)];
            float v = (dd > 0.0f) ? fabsf(g) * rsqrtf(dd) : 0.0f;
            if (g != g || dd != dd) v = __builtin_nanf("");
            loc = (v != v) ? v : ((loc != loc) ? loc : fmaxf(loc, v));
            const bool lead = ((i < 32) ? topI : topJ) || ((j < 32) ? topI : topJ);
            if (lead) {
                // termination measure = rotation ANGLE scale |g_ij| / max(g_ii, g_jj) = cos * sqrt(min/max): a tiny column
                // may keep a large cosine against a big one for many sweeps while the rotation it induces on the big
                // column (and the error it leaves in its sigma / vector) is already negligible.  This gives the small
                // singular values absolute accuracy eps*sigma_max (what LAPACK's bidiagonal SVD gives too) instead of
                // chasing their relative accuracy for 3-4 extra sweeps.
                const float mx = fmaxf(sdiag[0][i], sdiag[0][j]);
                float vt = (mx > 0.0f) ? fabsf(g) / mx : 0.0f;
                if (v != v) vt = v;
                loct = (vt != vt) ? vt : ((loct != loct) ? loct : fmaxf(loct, vt));
            }
        }
    }
    {
        float v = loc, vt = loct;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float u = __shfl_xor(v, o, 64);
            v = (u != u) ? u : ((v != v) ? v : fmaxf(v, u));
            const float ut = __shfl_xor(vt, o, 64);
            vt = (ut != ut) ? ut : ((vt != vt) ? vt : fmaxf(vt, ut));
        }
        if ((tid & 63) == 0) { redmax[tid >> 6] = v; redmax_t[tid >> 6] = vt; }
    }
    __syncthreads();
    float off0 = redmax[0], offt = redmax_t[0];
    for (int i = 1; i < 4; ++i) {
        const float u = redmax[i];
        off0 = (u != u) ? u : ((off0 != off0) ? off0 : fmaxf(off0, u));
        const float ut = redmax_t[i];
        offt = (ut != ut) ? ut : ((offt != offt) ? offt : fmaxf(offt, ut));
    }
    const bool is_nan = (off0 != off0);
    if (hist && tid == 0 && !is_nan) {  // debug: decade histogram of the pair measure (ASVD_DEBUG_HIST)
        int bk = (off0 > 0.0f) ? (int)floorf(-log10f(off0)) : 9;
        atomicAdd(&hist[bk < 0 ? 0 : (bk > 9 ? 9 : bk)], 1);
    }
    if (tid == 0) atomicMax(&maxoff_bits[b], is_nan ? 0x7fc00000u : __float_as_uint(offt));
    const bool rotate = !(is_nan || off0 < tol);
    if (tid == 0) {
        *act_flag = rotate ? 1 : 0;
        if (rotate && offt >= tol) atomicAdd(&nrot[b], 1);
    }
    if constexpr (MODE == 0) {
        if (!rotate) {
            if (v3.Gd32) {  // carried diagonal blocks of the two panels = the blocks of the matrix itself
                float* d0 = v3.Gd32 + ((int64_t)b * v3.nbpan + I) * 1024;
                float* d1 = v3.Gd32 + ((int64_t)b * v3.nbpan + J) * 1024;
                for (int e = tid; e < 1024; e += 256) {
                    const int i = e >> 5, j = e & 31;
                    d0[e] = G[i * PW + pcol(j)];
                    d1[e] = G[(32 + i) * PW + pcol(32 + j)];
                }
            }
            ASVD_KERNEL_RELEASE(sc);
            return;
        }
    }

    const int g = tid >> 5, tx = tid & 31;
    const int half_base = tid & 32;  // first lane of this half-wave within the wave
    // a nearly diagonal pair needs one sweep (quadratic convergence finishes the job at the next visit)
    const int nsw = (off0 > 0.05f) ? inner_sweeps : min(1, inner_sweeps);
    float qa[8], qb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        qa[r] = (8 * g + r == 2 * tx) ? 1.0f : 0.0f;
        qb[r] = (8 * g + r == 2 * tx + 1) ? 1.0f : 0.0f;
    }
    int cur = 0;
    // one phase, parity known at compile time (positions, idle tests and the DPP pattern constant-fold per parity)
    auto phase = [&](auto parc) {
        constexpr int par = decltype(parc)::value;
        // positions of this lane's column pair; phase B pair 31 = (63, 0) is idle (identity, no swap)
        const int cp = par ? ((2 * tx + 1) & 63) : 2 * tx;
        const int cq = par ? ((2 * tx + 2) & 63) : 2 * tx + 1;
        const bool col_idle = par && tx == 31;
        float c, s, t;
        const float da = sdiag[cur][cp], dd = sdiag[cur][cq], bb = sb[cur][tx];
        jacobi_rot(da, dd, bb, c, s, t);
        if (col_idle) { c = 1.0f; s = 0.0f; t = 0.0f; }
        // column coefficients (rotate + swap):  new[cp] = al*x[cp] + be*x[cq] ; new[cq] = ga*x[cp] + de*x[cq]
        const float al = col_idle ? 1.0f : s, be = col_idle ? 0.0f : c, ga = col_idle ? 0.0f : c, de = col_idle ? 1.0f : -s;
        const int acp = pcol(cp), acq = pcol(cq);
        const int nxt = cur ^ 1;
        float piv_p = 0.0f, piv_q = 0.0f, nb_val = 0.0f;  // new diagonal of my pivot block, next pivot off-diagonal I produce
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = 4 * g + u;  // row pair index
            const int rp = par ? ((2 * a + 1) & 63) : 2 * a;
            const int rq = par ? ((2 * a + 2) & 63) : 2 * a + 1;
            const bool row_idle = par && a == 31;
            const float rc = __shfl(c, half_base + a, 64), rs = __shfl(s, half_base + a, 64);
            const float ral = row_idle ? 1.0f : rs, rbe = row_idle ? 0.0f : rc, rga = row_idle ? 0.0f : rc, rde = row_idle ? 1.0f : -rs;
            const float x00 = G[rp * PW + acp], x01 = G[rp * PW + acq];
            const float x10 = G[rq * PW + acp], x11 = G[rq * PW + acq];
            // rows: new[rp] = ral*row[rp] + rbe*row[rq] ; new[rq] = rga*row[rp] + rde*row[rq]
            const float y00 = ral * x00 + rbe * x10, y01 = ral * x01 + rbe * x11;
            const float y10 = rga * x00 + rde * x10, y11 = rga * x01 + rde * x11;
            float z00 = al * y00 + be * y01, z01 = ga * y00 + de * y01;
            float z10 = al * y10 + be * y11, z11 = ga * y10 + de * y11;
            // pivot block: exact update, annihilated off-diagonal, swapped diagonal (position p now holds the rotated q:
            // d' = d + t b; position q holds a' = a - t b)
            const bool pivot = (a == tx) && !col_idle;
            z00 = pivot ? dd + t * bb : z00;
            z11 = pivot ? da - t * bb : z11;
            z01 = pivot ? 0.0f : z01;
            z10 = pivot ? 0.0f : z10;
            piv_p = pivot ? z00 : piv_p;
            piv_q = pivot ? z11 : piv_q;
            nb_val = (tx == ((a + 1) & 31)) ? z10 : nb_val;
            G[rp * PW + acp] = z00;
            G[rp * PW + acq] = z01;
            G[rq * PW + acp] = z10;
            G[rq * PW + acq] = z11;
        }
        // side arrays for the next phase (one predicated region per thread instead of one per block):
        //   diagonal from the pivot block (row pair tx lives in thread group tx/4);
        //   next pivot off-diagonal = element (rq, cp) of the block whose column pair is (row pair + 1) mod 32:
        //     after phase A the next pivots are the B pairs k = a (positions 2a+1, 2a+2), k <= 30,
        //     after phase B the A pairs k = a + 1 mod 32 (positions 2k, 2k+1).
        if ((tx >> 2) == g && !col_idle) {
            sdiag[nxt][cp] = piv_p;
            sdiag[nxt][cq] = piv_q;
        }
        {
            const int a_nb = (tx + 31) & 31;  // the row pair a with tx == a + 1 (mod 32)
            if ((a_nb >> 2) == g) {
                if (par == 0) { if (a_nb < 31) sb[nxt][a_nb] = nb_val; }
                else sb[nxt][tx] = nb_val;
            }
        }
        if (par && tid == 0) {  // idle positions keep their diagonal; idle B pair has no pivot
            sdiag[nxt][63] = sdiag[cur][63];
            sdiag[nxt][0] = sdiag[cur][0];
        }
        // eigenvector accumulation in REGISTERS: lane tx keeps rows 8g..8g+7 of the columns at positions 2tx (qa) and 2tx+1 (qb).
        // Phase A rotates (qa, qb) in place.  Phase B pairs positions (2tx+1, 2tx+2): qb with the qa of lane tx+1, fetched and
        // handed back with DPP wave shifts (VALU data path: the LDS, which bounds this kernel, is left to G alone).
        if constexpr (par == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float u0 = qa[r], v0 = qb[r];
                qa[r] = al * u0 + be * v0;
                qb[r] = ga * u0 + de * v0;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(qa[r]), 0x130 /*wave_shl:1: from lane+1*/, 0xF, 0xF, false));
                const float u0 = qb[r];
                qb[r] = al * u0 + be * v0;                 // position 2tx+1
                const float back = ga * u0 + de * v0;      // position 2tx+2 -> lane tx+1's qa (identity on the idle lane 31)
                const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(back), 0x138 /*wave_shr:1: from lane-1*/, 0xF, 0xF, false));
                qa[r] = (tx == 0) ? qa[r] : recv;          // position 0 is idle in phase B
            }
        }
        __syncthreads();
        cur = nxt;
    };
    for (int ph2 = 0; ph2 < (rotate ? nsw * sc.evd_pairs : 0); ++ph2) {
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
    }
    // eigenvectors to LDS (plane-major image) for the normalisation / sort / store epilogue
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        Q[(8 * g + r) * PW + pcol(2 * tx)] = qa[r];
        Q[(8 * g + r) * PW + pcol(2 * tx + 1)] = qb[r];
    }
    __syncthreads();

    // column norms of Q in double (4 threads x 16 rows per column): Q's columns are renormalised to unit length so
    // that the accumulated rounding of ~64-128 rotations per column cannot drift the norms of the updated panels.
    {
        const int cpos = tid >> 2, part = tid & 3;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double v = Q[(part * 16 + r) * PW + pcol(cpos)];
            acc += v * v;
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (part == 0) cscale[cpos] = (rotate && acc > 0.0) ? (float)(1.0 / sqrt(acc)) : 1.0f;
    }
    // sort eigenvalues descending (ties by index): the column at position c goes to output column rnk[c]; nothing moves when the
    // solve did not rotate
    if (tid < PW) {
        const float me = sdiag[cur][tid];
        int cnt = 0;
        for (int i = 0; i < PW; ++i) {
            const float o = sdiag[cur][i];
            cnt += (o > me || (o == me && i < tid)) ? 1 : 0;
        }
        rnk[tid] = rotate ? cnt : tid;
    }
    __syncthreads();
    if constexpr (MODE <= 1) {
        for (int e = tid; e < PW * PW; e += 256) {
            const int r = e >> 6, c = e & 63;
            qo[r * PW + rnk[c]] = Q[r * PW + pcol(c)] * cscale[c];
        }
    }
    if constexpr (MODE == 0) {
        if (v3.Gd32)  // Q^T G Q (what the rotations left in LDS), in the order and scaling of Q's columns
            store_diag_blocks(G, rnk, cscale, v3.Gd32 + ((int64_t)b * v3.nbpan + I) * 1024, v3.Gd32 + ((int64_t)b * v3.nbpan + J) * 1024, tid);
    } else if constexpr (MODE == 1) {
        const int ba = sp, bb = 2 + sp;
        store_diag_blocks(G, rnk, cscale, v3.D0 + (slot * 4 + ba) * 1024, v3.D0 + (slot * 4 + bb) * 1024, tid);
    } else {
        const int ba = sp, bb = 3 - sp;
        store_diag_blocks(G, rnk, cscale, v3.Gd32 + ((int64_t)b * v3.nbpan + I) * 1024, v3.Gd32 + ((int64_t)b * v3.nbpan + J) * 1024, tid);
        __syncthreads();  // G consumed
        // Qfin[:, columns of blocks (a, b)] = Q^(0)[:, {a, b}] Q1:  rows of blocks {a, a+2} get Q0_a[:, :32] Q1[:32, :], rows of blocks
        // {b-2, b} get Q0_(b-2)[:, 32:] Q1[32:, :].  Q1 (sorted, rescaled) goes row-major into G's LDS, the two Q0 halves TRANSPOSED
        // (k-major, so that the MFMA A operand reads are contiguous) over Qs once Q has been consumed.
        float* Q1s = G;
        for (int e = tid; e < PW * PW; e += 256) {
            const int r = e >> 6, c = e & 63;
            Q1s[r * PW + rnk[c]] = Q[r * PW + pcol(c)] * cscale[c];
        }
        __syncthreads();
        float* At0 = Qs;          // At0[k][i] = Q0_a[i][k],        k < 32, i < 64
        float* At1 = Qs + 2048;   // At1[k][i] = Q0_(b-2)[i][32 + k]
        const float* __restrict__ q0a = v3.Q0 + (slot * 2 + ba) * (PW * PW);
        const float* __restrict__ q0b = v3.Q0 + (slot * 2 + (bb - 2)) * (PW * PW);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q, i = e >> 5, k = e & 31;
            At0[k * PW + i] = q0a[i * PW + k];
            At1[k * PW + i] = q0b[i * PW + 32 + k];
        }
        __syncthreads();
        const int hh = lane >> 5, cc = lane & 31;
        float* __restrict__ qf = v3.Qfin + slot * (128 * 128);
        // 8 output tiles (2 products x 2 x 2 tiles of 32 x 32, K = 32): wave wv does product wv >> 1, row tile wv & 1, both column tiles
        const int prod = wv >> 1, ti = wv & 1;
        const float* At = prod ? At1 : At0;
        const float* Bm = Q1s + (prod ? 32 * PW : 0);
        f32x16 c0 = {0}, c1 = {0};
#pragma unroll 8
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = 2 * k2 + hh;
            const float a = At[k * PW + 32 * ti + cc];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bm[k * PW + cc], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bm[k * PW + 32 + cc], c1, 0, 0, 0);
        }
        // row block of local row tile ti: product 0 -> blocks {a, a+2}[ti]; product 1 -> blocks {b-2, b}[ti]
        const int rblk = prod ? (ti ? bb : bb - 2) : (ti ? ba + 2 : ba);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            qf[(32 * rblk + i) * 128 + 32 * ba + cc] = c0[reg];
            qf[(32 * rblk + i) * 128 + 32 * bb + cc] = c1[reg];
        }
    }
    ASVD_KERNEL_RELEASE(sc);
}

template <int MODE, int KEEPG>
__global__ __launch_bounds__(256, 4) void evd_kernel(Sched sc, const float* __restrict__ Gpart, int nsplit, float* __restrict__ Qbuf,
                                                   int* __restrict__ active, unsigned* __restrict__ maxoff_bits,
                                                   int* __restrict__ nrot, const int* __restrict__ done, float tol,
                                                   int inner_sweeps, int nb, int step, int kb, int* __restrict__ hist,
                                                   const int* __restrict__ plist, int list_stride, EvdV3 v3) {
    __shared__ __attribute__((aligned(16))) float smem[EVD_SMEM_FLOATS(KEEPG)];
    const BlockCtx ctx{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z};
    evd_body<MODE, KEEPG>(sc, ctx, smem, Gpart, nsplit, Qbuf, active, maxoff_bits, nrot, done, tol, inner_sweeps, nb, step, kb, hist, plist,
                          list_stride, v3);
}

// --------------------------------------------------------------------------------------------------
// update: [X_I X_J] <- [X_I X_J] * Q for the rows of this chunk.  Each wave owns 32-row tiles: the tile is
// staged through a private padded LDS image (coalesced 1-KB global loads in, conflict-free ds_read_b128
// row-per-lane out), multiplied by Q held in 64 VGPRs, and stored as full 128-B row segments.
constexpr int TLD = PW + 4;  // LDS row stride in floats (272 B, multiple of 16 B; bank-conflict-free b128 reads)

__global__ __launch_bounds__(256) void update_kernel(Sched sc, float* __restrict__ X, int64_t panel_stride, int64_t batch_stride,
                                                     int nb, int step, int R, int rows_per_wg,
                                                     const float* __restrict__ Qbuf, const int* __restrict__ active,
                                                     const int* __restrict__ done, const int* __restrict__ plist, int list_stride) {
    const int chunk = blockIdx.x, pair = blockIdx.y, b = blockIdx.z, npairs = gridDim.y;
    ASVD_KERNEL_ACQUIRE(sc);
    if (ld_flag(done + b) || !ld_flag(active + b * npairs + pair)) return;
    int I, J;
    if (!get_pair(sc, plist, list_stride, b, nb, step, pair, I, J)) return;
    float* __restrict__ XI = X + (int64_t)b * batch_stride + (int64_t)I * panel_stride;
    float* __restrict__ XJ = X + (int64_t)b * batch_stride + (int64_t)J * panel_stride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;

    const float* __restrict__ Qp = Qbuf + ((int64_t)b * npairs + pair) * (PW * PW);
    float q0[32], q1[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        q0[t] = Qp[(h * 32 + t) * PW + c];
        q1[t] = Qp[(h * 32 + t) * PW + 32 + c];
    }

    __shared__ __attribute__((aligned(16))) float tile[4][32 * TLD];
    float* my = tile[w];
    const int r_begin = chunk * rows_per_wg;
    const int r_end = min(r_begin + rows_per_wg, R);
    // a wave walks its 32-row tiles with a stride of 128 rows.  The LDS tile is wave-private and a wave's LDS operations complete in
    // order, so the loop needs no workgroup barrier.  (Prefetching the next tile into registers was measured: 224 -> 232 us per
    // launch — it costs the third wave per SIMD.)
    for (int r0 = r_begin + w * 32; r0 < r_end; r0 += 128) {  // R and rows_per_wg are multiples of 32
        {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 256 + lane * 4;
                const int row = idx >> 5, col = idx & 31;
                const f32x4 vi = *(const f32x4*)(XI + (int64_t)r0 * PB + idx);
                const f32x4 vj = *(const f32x4*)(XJ + (int64_t)r0 * PB + idx);
                *(f32x4*)(my + row * TLD + col) = vi;
                *(f32x4*)(my + row * TLD + 32 + col) = vj;
            }
            float a[32];
#pragma unroll
            for (int t4 = 0; t4 < 8; ++t4) {
                const f32x4 v = *(const f32x4*)(my + c * TLD + h * 32 + t4 * 4);
                a[4 * t4 + 0] = v[0];
                a[4 * t4 + 1] = v[1];
                a[4 * t4 + 2] = v[2];
                a[4 * t4 + 3] = v[3];
            }
            f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], q0[t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], q1[t], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                XI[(int64_t)(r0 + i) * PB + c] = acc0[reg];
                XJ[(int64_t)(r0 + i) * PB + c] = acc1[reg];
            }
        }
    }
    ASVD_KERNEL_RELEASE(sc);
}

// bit surgery for the XOR schedule: pair slot <-> lower member, quad index -> representative
__device__ __forceinline__ int insert_zero_bit(int v, int pos) { return ((v >> pos) << (pos + 1)) | (v & ((1 << pos) - 1)); }
__device__ __forceinline__ int remove_bit(int v, int pos) { return ((v >> (pos + 1)) << pos) | (v & ((1 << pos) - 1)); }
#include "twolevel.h"

// --------------------------------------------------------------------------------------------------
// Sparse sweeps.  Once fewer than half of the pairs still rotate, most of a sweep is Gram passes that only confirm convergence
// (the last sweep of a 4096^2 problem rotates 0.3 % of its pairs and still costs a third of a full sweep, all of it panel reads).
// A sparse sweep starts with ONE snapshot of all couplings — X^T X as a blocked GEMM: each panel is read nb/4 times through L2
// instead of nb-1 times from HBM and the diagonal blocks are not recomputed per pair — which marks the pairs whose scaled
// coupling is >= tol.  The host turns the marks into per-step lists (XOR steps are perfect matchings, so the pairs of one step
// are disjoint) and launches gram / evd / update for marked pairs only, skipping empty steps.  Couplings of unmarked pairs move
// only by (rotation angle) x (other couplings) during the sweep, second order in what is left; the next snapshot sees them.
// The termination measure is the snapshot's (same definition as in evd_kernel), so the stopping rule is unchanged.
__global__ __launch_bounds__(256) void panel_sumsq_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride,
                                                          int m_pad, int n_pad, float* __restrict__ dn, const int* __restrict__ done) {
    const int I = blockIdx.x, b = blockIdx.y, c = threadIdx.x & 31, g = threadIdx.x >> 5;
    ASVD_KERNEL_ACQUIRE(sc);
    if (ld_flag(done + b)) return;
    const float* __restrict__ P = X + (int64_t)b * batch_stride + (int64_t)I * panel_stride;
    float s = 0.0f;
    for (int r = g; r < m_pad; r += 8) { const float x = P[(int64_t)r * PB + c]; s = fmaf(x, x, s); }
    __shared__ float red[8][32];
    red[g][c] = s;
    __syncthreads();
    if (g == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][c];
        dn[(int64_t)b * n_pad + I * PB + c] = t;
    }
    ASVD_KERNEL_RELEASE(sc);
}

__device__ __forceinline__ float nanmax(float a, float b) { return (b != b) ? b : ((a != a) ? a : fmaxf(a, b)); }

// grid (ceil(nb/4), ceil(nb/4), batch); upper-triangular tiles only.  Wave w owns panel J = 4*jg + w against panels I = 4*ig + a.
// 32-row chunks of the eight panels are staged in LDS (coalesced 16-B loads, next chunk prefetched into registers while the
// current one is in the matrix pipe); every wave reads its operands from LDS as conflict-free 256-B rows.
template <int SPLIT>
__global__ __launch_bounds__(256) void fullcheck_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int nb,
                                                        int m_pad, int n_pad, const float* __restrict__ dn, float tol, int kb,
                                                        unsigned char* __restrict__ pflag, unsigned* __restrict__ maxoff_bits,
                                                        const int* __restrict__ done) {
    // (An XCD-aware tile order — the workgroups of one XCD walking a contiguous run of tiles — was measured SLOWER: 66.8 vs 55.6 ms for
    // the three snapshots of 32 problems; the plain order stays.)
    const int ig = blockIdx.x, jg = blockIdx.y, b = blockIdx.z;
    if (ig > jg || ld_flag(done + b)) return;
    ASVD_KERNEL_ACQUIRE(sc);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int J = jg * 4 + w;
    bool ok[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) ok[a] = (ig * 4 + a <= J) && (J < nb);  // blocks below the diagonal are mirrors
    // slot q of the stage: q < 4 -> panel 4*ig + q (A side), q >= 4 -> panel 4*jg + q - 4 (B side); clamp padding panels
    const float* __restrict__ Xb = X + (int64_t)b * batch_stride;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (f32x16){0};
    if constexpr (SPLIT) {
        // split-bf16 (twolevel.h): each fp32 operand = three bf16 exactly, six products per fp32 product on the bf16 matrix pipe (2.7x less
        // pipe time; the dropped terms are at fp32 rounding level, which a coupling test against tol = 1e-6 needs).  The operand of k-step
        // ks of a panel is: lane (column cc, group hh) holds rows 16 ks + 8 hh + e, e = 0..7.  Every operand of a 32-row chunk is built
        // ONCE per workgroup: thread t owns the slots (panel (t >> 7) + 2 j, k-step (t >> 6) & 1, lane t & 63), j = 0..3, loads its eight
        // values straight from global memory (a wave-load covers two 128-byte row segments), splits them and stores the three parts as
        // ready operands; the waves then only read 16-byte operands.  (Before: every wave split the four A panels and its own B panel
        // itself from an fp32 LDS image — 40 splits and 80 scalar LDS reads per wave and chunk against 48 MFMAs.)
        __shared__ u32x4 oimg[8 * 2 * 3 * 64];
        const int sks = (tid >> 6) & 1, sl = tid & 63;
        const float* src[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = (tid >> 7) + 2 * j;
            const int pnl = (q < 4) ? ig * 4 + q : jg * 4 + q - 4;
            src[j] = Xb + (int64_t)(pnl < nb ? pnl : nb - 1) * panel_stride + (int64_t)(16 * sks + 8 * (sl >> 5)) * PB + (sl & 31);
        }
        float pre[4][8];
        auto fetch = [&](int r0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) pre[j][e] = src[j][(int64_t)(r0 + e) * PB];
        };
        fetch(0);
        for (int r0 = 0; r0 < m_pad; r0 += 32) {
            __syncthreads();  // previous chunk's operands fully consumed
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 p1, p2, p3;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned x, y, z;
                    split3(pre[j][2 * e2], pre[j][2 * e2 + 1], x, y, z);
                    p1[e2] = x; p2[e2] = y; p3[e2] = z;
                }
                u32x4* o = oimg + ((((tid >> 7) + 2 * j) * 2 + sks) * 3) * 64 + sl;
                o[0] = p1; o[64] = p2; o[128] = p3;
            }
            __syncthreads();
            if (r0 + 32 < m_pad) fetch(r0 + 32);
            // eight stages (k-step, A panel) of six MFMAs; the operands of stage i + 1 are read from LDS before the MFMAs of stage i are
            // issued (left to the compiler each ds_read sat in front of its consumer: 51 % of the wave cycles waiting to issue)
            struct Op3 { bf16x8 p1, p2, p3; };
            auto ld3 = [&](int slot, int ks) {
                const u32x4* o = oimg + ((slot * 2 + ks) * 3) * 64 + lane;
                Op3 r;
                r.p1 = __builtin_bit_cast(bf16x8, o[0]); r.p2 = __builtin_bit_cast(bf16x8, o[64]); r.p3 = __builtin_bit_cast(bf16x8, o[128]);
                return r;
            };
            Op3 Bc = ld3(4 + w, 0), Ac = ld3(0, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int ks = st >> 2, a = st & 3;
                Op3 An = Ac, Bn = Bc;
                if (st + 1 < 8) An = ld3((st + 1) & 3, (st + 1) >> 2);
                if (st == 3) Bn = ld3(4 + w, 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.p3, Bc.p1, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.p1, Bc.p3, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.p2, Bc.p2, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.p2, Bc.p1, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.p1, Bc.p2, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.p1, Bc.p1, acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                Ac = An;
                if (st == 3) Bc = Bn;
                (void)ks;
            }
        }
    } else {
        __shared__ __attribute__((aligned(16))) float stage[8][32 * PB];
        const float* src[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pnl = (q < 4) ? ig * 4 + q : jg * 4 + q - 4;
            src[q] = Xb + (int64_t)(pnl < nb ? pnl : nb - 1) * panel_stride + tid * 4;  // 256 threads x 16 B = one 32x32 chunk
        }
        f32x4 pre[8];
        auto fetch = [&](int r0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) pre[q] = *(const f32x4*)(src[q] + (int64_t)r0 * PB);
        };
        fetch(0);
        for (int r0 = 0; r0 < m_pad; r0 += 32) {
            __syncthreads();  // previous chunk fully consumed
#pragma unroll
            for (int q = 0; q < 8; ++q) *(f32x4*)(&stage[q][tid * 4]) = pre[q];
            __syncthreads();
            if (r0 + 32 < m_pad) fetch(r0 + 32);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float bf = stage[4 + w][u * 64 + lane];
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(stage[a][u * 64 + lane], bf, acc[a], 0, 0, 0);
            }
        }
    }
    if (J >= nb) { ASVD_KERNEL_RELEASE(sc); return; }
    const int h = lane >> 5, c = lane & 31;
    const float* __restrict__ dnb = dn + (int64_t)b * n_pad;
    const float dj = dnb[J * PB + c];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (!ok[a]) continue;
        const int I = ig * 4 + a;
        float v = 0.0f, vt = 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (I == J && i == c) continue;
            const float di = dnb[I * PB + i];
            const float g = acc[a][reg];
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
    ASVD_KERNEL_RELEASE(sc);
}

// One 32x32 block of the snapshot: scaled couplings of panel I's columns against panel J's, mark + termination measure.
__device__ __forceinline__ void fullcheck_block(const f32x16& acc, int I, int J, int lane, const float* __restrict__ dnb, float dj, float tol, int kb,
                                                int nb, int64_t b, unsigned char* __restrict__ pflag, unsigned* __restrict__ maxoff_bits) {
    const int h = lane >> 5, c = lane & 31;
    float v = 0.0f, vt = 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (I == J && i == c) continue;
        const float di = dnb[I * PB + i];
        const float g = acc[reg];
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
                pflag[(b * nb + A) * nb + Bp] = 1;
            }
            if (I < kb || J < kb) atomicMax(&maxoff_bits[b], __float_as_uint(vt));
        }
    }
}

// The same snapshot on 256x256 tiles: grid (ceil(nb/8), ceil(nb/8), batch), 512 threads, wave w owns panel J = 8*jg + w against the
// eight panels I = 8*ig + a.  fullcheck_kernel<1> re-reads every panel nb/4 times and was bound by that traffic (39 GB per launch
// measured at the fabric for 2 GB of panels, round 2); doubling the tile edge halves it at the same matrix-pipe work.  The sixteen
// operand images of a 32-row chunk take 96 KB of LDS, one workgroup (two waves per SIMD) per CU.
constexpr int FC8_LDS_BYTES = 16 * 2 * 3 * 64 * 16;
__global__ __launch_bounds__(512, 1) void fullcheck8_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int nb,
                                                            int m_pad, int n_pad, const float* __restrict__ dn, float tol, int kb,
                                                            unsigned char* __restrict__ pflag, unsigned* __restrict__ maxoff_bits,
                                                            const int* __restrict__ done) {
    const int ig = blockIdx.x, jg = blockIdx.y, b = blockIdx.z;
    if (ig > jg || ld_flag(done + b)) return;
    ASVD_KERNEL_ACQUIRE(sc);
    extern __shared__ __attribute__((aligned(16))) u32x4 fc8_oimg[];  // [16 panels][2 k-steps][3 parts][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int J = jg * 8 + w;
    const float* __restrict__ Xb = X + (int64_t)b * batch_stride;
    f32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = (f32x16){0};
    // thread t builds the operands (panel slot (t >> 7) + 4 j, k-step (t >> 6) & 1, lane t & 63), j = 0..3; slots 0..7 = A side, 8..15 = B side
    const int sks = (tid >> 6) & 1, sl = tid & 63, q0 = tid >> 7;
    const float* src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + 4 * j;
        const int pnl = (q < 8) ? ig * 8 + q : jg * 8 + q - 8;
        src[j] = Xb + (int64_t)(pnl < nb ? pnl : nb - 1) * panel_stride + (int64_t)(16 * sks + 8 * (sl >> 5)) * PB + (sl & 31);
    }
    float pre[4][8];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) pre[j][e] = src[j][(int64_t)(r0 + e) * PB];
    };
    fetch(0);
    for (int r0 = 0; r0 < m_pad; r0 += 32) {
        __syncthreads();  // previous chunk's operands fully consumed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 p1, p2, p3;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                unsigned x, y, z;
                split3(pre[j][2 * e2], pre[j][2 * e2 + 1], x, y, z);
                p1[e2] = x; p2[e2] = y; p3[e2] = z;
            }
            u32x4* o = fc8_oimg + (((q0 + 4 * j) * 2 + sks) * 3) * 64 + sl;
            o[0] = p1; o[64] = p2; o[128] = p3;
        }
        __syncthreads();
        if (r0 + 32 < m_pad) fetch(r0 + 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4* ob = fc8_oimg + (((8 + w) * 2 + ks) * 3) * 64 + lane;
            const bf16x8 B1 = __builtin_bit_cast(bf16x8, ob[0]), B2 = __builtin_bit_cast(bf16x8, ob[64]), B3 = __builtin_bit_cast(bf16x8, ob[128]);
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const u32x4* oa = fc8_oimg + ((a * 2 + ks) * 3) * 64 + lane;
                const bf16x8 A1 = __builtin_bit_cast(bf16x8, oa[0]), A2 = __builtin_bit_cast(bf16x8, oa[64]), A3 = __builtin_bit_cast(bf16x8, oa[128]);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[a], 0, 0, 0);
            }
        }
    }
    if (J >= nb) { ASVD_KERNEL_RELEASE(sc); return; }
    const float* __restrict__ dnb = dn + (int64_t)b * n_pad;
    const float dj = dnb[J * PB + (lane & 31)];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int I = ig * 8 + a;
        if (I > J) continue;  // blocks below the diagonal are mirrors
        fullcheck_block(acc[a], I, J, lane, dnb, dj, tol, kb, nb, b, pflag, maxoff_bits);
    }
    ASVD_KERNEL_RELEASE(sc);
}

// --------------------------------------------------------------------------------------------------
// upgram: update of step d fused with the Gram matrices of step e (the step that follows).  Under the XOR ordering the four
// panels {a, a^d, a^e, a^d^e} are closed under both steps: pairs (a, a^d), (a^e, a^d^e) rotate now, pairs (a, a^e), (a^d, a^d^e)
// meet next.  One workgroup streams a row chunk of such a quad ONCE: rotate both pairs (fp32 MFMA against the 64x64 Q held in
// registers), write the panels back, and accumulate the six 32x32 Gram blocks of the two next pairs from the updated tile in
// LDS.  HBM traffic per step drops from 3 panel passes (gram read + update read + update write) to 2; the kernel is MFMA-bound
// (224 MFMA per 32-row tile per wave), so it runs one workgroup per CU with the whole register file (Q 128 + accumulators 128
// + prefetch 64 VGPRs) and prefetches the next tile into registers while the current one is in the matrix pipe.
constexpr int TLQ = 4 * PB + 4;  // LDS row stride of the quad tile in floats (528 B: 16-B multiple, conflict-free b128 row reads)


__global__ __launch_bounds__(256, 1) void upgram_kernel(float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int nb,
                                                        int d, int e, int R, int m_pad, int rows_per_wg,
                                                        const float* __restrict__ Qbuf, const int* __restrict__ active,
                                                        float* __restrict__ Gpart, const int* __restrict__ done) {
    const int chunk = blockIdx.x, quad = blockIdx.y, b = blockIdx.z, nsplit = gridDim.x, npairs = nb >> 1;
    if (ld_flag(done + b)) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    // ---- quad geometry (uniform per workgroup) ----
    const int h1 = 31 - __clz(d);
    const int e1 = ((e >> h1) & 1) ? (e ^ d) : e;
    const int h2 = 31 - __clz(e1);
    const int lo = min(h1, h2), hi = max(h1, h2);
    const int a0 = insert_zero_bit(insert_zero_bit(quad, lo), hi);
    const int P0 = a0, P1 = a0 ^ d, P2 = a0 ^ e, P3 = a0 ^ d ^ e;  // tile slots 0..3
    // current pairs (step d): A = slots (0,1), B = slots (2,3); the lower-index member has bit h1 clear
    const bool swapB = (P2 >> h1) & 1;
    const int kA = remove_bit(P0, h1), kB = remove_bit(swapB ? P3 : P2, h1);
    const bool actA = ld_flag(active + b * npairs + kA) != 0, actB = ld_flag(active + b * npairs + kB) != 0;
    // next pairs (step e): C = slots (0,2), D = slots (1,3)
    const int he = 31 - __clz(e);
    const bool swapD = (P1 >> he) & 1;
    const int kC = remove_bit(P0, he), kD = remove_bit(swapD ? P3 : P1, he);

    float* __restrict__ Xb = X + (int64_t)b * batch_stride;
    float* __restrict__ Xp[4] = {Xb + (int64_t)P0 * panel_stride, Xb + (int64_t)P1 * panel_stride, Xb + (int64_t)P2 * panel_stride,
                                 Xb + (int64_t)P3 * panel_stride};
    // Q of the two current pairs: rows (h*32 + t), columns c (first output panel) and 32 + c (second)
    float qa0[32], qa1[32], qb0[32], qb1[32];
    if (actA) {
        const float* __restrict__ Qp = Qbuf + ((int64_t)b * npairs + kA) * (PW * PW);
#pragma unroll
        for (int t = 0; t < 32; ++t) { qa0[t] = Qp[(h * 32 + t) * PW + c]; qa1[t] = Qp[(h * 32 + t) * PW + 32 + c]; }
    }
    if (actB) {
        const float* __restrict__ Qp = Qbuf + ((int64_t)b * npairs + kB) * (PW * PW);
#pragma unroll
        for (int t = 0; t < 32; ++t) { qb0[t] = Qp[(h * 32 + t) * PW + c]; qb1[t] = Qp[(h * 32 + t) * PW + 32 + c]; }
    }
    const int sAi = 0, sAj = 1, sBi = swapB ? 3 : 2, sBj = swapB ? 2 : 3;  // tile slots of (I, J) of the current pairs

    __shared__ __attribute__((aligned(16))) float tile[4][32 * TLQ];
    float* my = tile[w];
    f32x16 gC0 = {0}, gC1 = {0}, gC2 = {0}, gD0 = {0}, gD1 = {0}, gD2 = {0};  // (II, IJ, JJ) of the next pairs

    const int r_begin = chunk * rows_per_wg;
    const int r_end = min(r_begin + rows_per_wg, R);
    f32x4 pre[4][4];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4)
#pragma unroll
            for (int it = 0; it < 4; ++it) pre[p4][it] = *(const f32x4*)(Xp[p4] + (int64_t)r0 * PB + it * 256 + lane * 4);
    };
    int r0 = r_begin + w * 32;  // R and rows_per_wg are multiples of 32; a wave strides by 128 rows
    if (r0 < r_end) prefetch(r0);
    for (; r0 < r_end; r0 += 128) {
        // registers -> this wave's LDS tile (wave-private: LDS operations of one wave complete in order, no barrier needed)
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 256 + lane * 4;
                *(f32x4*)(my + (idx >> 5) * TLQ + p4 * 32 + (idx & 31)) = pre[p4][it];
            }
        if (r0 + 128 < r_end) prefetch(r0 + 128);
        // ---- rotate the two current pairs ----
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const bool act = pr == 0 ? actA : actB;
            if (!act) continue;
            const int si = pr == 0 ? sAi : sBi, sj = pr == 0 ? sAj : sBj;
            float av[32];
            const float* src = my + c * TLQ + (h ? sj : si) * 32;
#pragma unroll
            for (int t4 = 0; t4 < 8; ++t4) {
                const f32x4 v = *(const f32x4*)(src + t4 * 4);
                av[4 * t4 + 0] = v[0]; av[4 * t4 + 1] = v[1]; av[4 * t4 + 2] = v[2]; av[4 * t4 + 3] = v[3];
            }
            f32x16 acc0 = {0}, acc1 = {0};
            if (pr == 0) {
#pragma unroll
                for (int t = 0; t < 32; ++t) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], qa0[t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], qa1[t], acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 32; ++t) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], qb0[t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], qb1[t], acc1, 0, 0, 0);
                }
            }
            float* __restrict__ XI = Xp[si];
            float* __restrict__ XJ = Xp[sj];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                XI[(int64_t)(r0 + i) * PB + c] = acc0[reg];
                XJ[(int64_t)(r0 + i) * PB + c] = acc1[reg];
                my[i * TLQ + si * 32 + c] = acc0[reg];
                my[i * TLQ + sj * 32 + c] = acc1[reg];
            }
        }
        // ---- Gram blocks of the next pairs from the updated tile (rows of the matrix proper only, not accumulated V rows) ----
        if (r0 < m_pad) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float* rowp = my + (2 * u + h) * TLQ + c;
                const float x0 = rowp[0], x1 = rowp[32], x2 = rowp[64], x3 = rowp[96];
                const float di = swapD ? x3 : x1, dj = swapD ? x1 : x3;
                gC0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x0, gC0, 0, 0, 0);
                gC1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x2, gC1, 0, 0, 0);
                gC2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x2, x2, gC2, 0, 0, 0);
                gD0 = __builtin_amdgcn_mfma_f32_32x32x2f32(di, di, gD0, 0, 0, 0);
                gD1 = __builtin_amdgcn_mfma_f32_32x32x2f32(di, dj, gD1, 0, 0, 0);
                gD2 = __builtin_amdgcn_mfma_f32_32x32x2f32(dj, dj, gD2, 0, 0, 0);
            }
        }
    }

    // ---- cross-wave reduction in a fixed order ((w0 + w2) + (w1 + w3)) through the tile memory, then wave 0 stores ----
    __syncthreads();
    float* red = &tile[0][0];  // 4 * 32 * 132 floats = 16896 >= 2 * 96 * 64 = 12288
    auto put = [&](float* dst) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            dst[(0 + reg) * 64 + lane] = gC0[reg]; dst[(16 + reg) * 64 + lane] = gC1[reg]; dst[(32 + reg) * 64 + lane] = gC2[reg];
            dst[(48 + reg) * 64 + lane] = gD0[reg]; dst[(64 + reg) * 64 + lane] = gD1[reg]; dst[(80 + reg) * 64 + lane] = gD2[reg];
        }
    };
    auto add = [&](const float* src) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            gC0[reg] += src[(0 + reg) * 64 + lane]; gC1[reg] += src[(16 + reg) * 64 + lane]; gC2[reg] += src[(32 + reg) * 64 + lane];
            gD0[reg] += src[(48 + reg) * 64 + lane]; gD1[reg] += src[(64 + reg) * 64 + lane]; gD2[reg] += src[(80 + reg) * 64 + lane];
        }
    };
    if (w >= 2) put(red + (w - 2) * 96 * 64);
    __syncthreads();
    if (w < 2) add(red + w * 96 * 64);
    __syncthreads();
    if (w == 1) put(red);
    __syncthreads();
    if (w == 0) {
        add(red);
        float* outC = Gpart + (((int64_t)b * npairs + kC) * nsplit + chunk) * 3072;
        float* outD = Gpart + (((int64_t)b * npairs + kD) * nsplit + chunk) * 3072;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            outC[0 * 1024 + i * 32 + c] = gC0[reg];
            outC[1 * 1024 + i * 32 + c] = gC1[reg];
            outC[2 * 1024 + i * 32 + c] = gC2[reg];
            outD[0 * 1024 + i * 32 + c] = gD0[reg];
            outD[1 * 1024 + i * 32 + c] = gD1[reg];
            outD[2 * 1024 + i * 32 + c] = gD2[reg];
        }
    }
}

// --------------------------------------------------------------------------------------------------
// backsolve: right vectors without accumulating V during the sweeps.  After convergence X_J holds a_j = sigma_j u_j and
//   Xorig^T a_j = V Sigma U^T (sigma_j u_j) = sigma_j^2 v_j ,
// so the V rows of panel J are the cross-Gram blocks between the ORIGINAL packed panels and the final ones: the same MFMA
// operand pattern as gram_kernel (one coalesced 256-B load per panel per 2 rows).  A workgroup computes 4 (I) x 4 (J)
// blocks; wave w owns J panel w (B operand) against the 4 I panels (A operands, shared through L1 by the 4 waves).
// The finalize kernels normalise the rows block to unit columns, so the sigma_j^2 factor is irrelevant.
__global__ __launch_bounds__(256) void backsolve_kernel(const float* __restrict__ Xorig, int64_t orig_panel_stride,
                                                        int64_t orig_batch_stride, float* __restrict__ X, int64_t panel_stride,
                                                        int64_t batch_stride, int nb, int m_pad) {
    const int ig = blockIdx.x, jg = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int J = jg * 4 + w;
    if (J >= nb) return;  // no barriers below
    const float* __restrict__ pj = X + (int64_t)b * batch_stride + (int64_t)J * panel_stride + lane;
    const float* pi[4];
    bool ok[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int I = ig * 4 + a;
        ok[a] = I < nb;
        pi[a] = Xorig + (int64_t)b * orig_batch_stride + (int64_t)(ok[a] ? I : 0) * orig_panel_stride + lane;
    }
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (f32x16){0};
    const int nsteps = m_pad >> 1;
    int s = 0;
    for (; s + 4 <= nsteps; s += 4) {
        float bf[4], af[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t off = (int64_t)(s + u) * (2 * PB);
            bf[u] = pj[off];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[u][a] = pi[a][off];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u][a], bf[u], acc[a], 0, 0, 0);
    }
    for (; s < nsteps; ++s) {
        const int64_t off = (int64_t)s * (2 * PB);
        const float bfr = pj[off];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(pi[a][off], bfr, acc[a], 0, 0, 0);
    }
    const int h = lane >> 5, c = lane & 31;
    float* __restrict__ out = X + (int64_t)b * batch_stride + (int64_t)J * panel_stride;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (!ok[a]) continue;
        const int I = ig * 4 + a;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            out[(int64_t)(m_pad + I * PB + i) * PB + c] = acc[a][reg];
        }
    }
}

// --------------------------------------------------------------------------------------------------
// finalize 1: column norms (double accumulation), sigma_j = |a_j| / |v_j| (drift-corrected) or |a_j|
__global__ __launch_bounds__(256) void colnorm_kernel(const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride,
                                                      int m_pad, int R, int n_pad, int sig_ratio, float* __restrict__ sig,
                                                      float* __restrict__ inv_na, float* __restrict__ inv_nv) {
    const int blk = blockIdx.x, b = blockIdx.y;
    const float* P = X + (int64_t)b * batch_stride + (int64_t)blk * panel_stride;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;  // 8 row groups
    double sa = 0.0, sv = 0.0;
    for (int r = g; r < m_pad; r += 8) {
        const double v = P[(int64_t)r * PB + c];
        sa += v * v;
    }
    for (int r = m_pad + g; r < R; r += 8) {
        const double v = P[(int64_t)r * PB + c];
        sv += v * v;
    }
    __shared__ double ra[8][32], rv[8][32];
    ra[g][c] = sa;
    rv[g][c] = sv;
    __syncthreads();
    if (g == 0) {
        for (int i = 1; i < 8; ++i) { sa += ra[i][c]; sv += rv[i][c]; }
        const double na = sqrt(sa), nv = sqrt(sv);
        const int j = blk * PB + c;
        double s = na;
        if (sig_ratio) s = (nv > 0.0) ? na / nv : 0.0;  // accumulated V: drift-corrected sigma = |a_j| / |v_j|
        sig[(int64_t)b * n_pad + j] = (float)s;
        inv_na[(int64_t)b * n_pad + j] = (na > 0.0) ? (float)(1.0 / na) : 0.0f;
        inv_nv[(int64_t)b * n_pad + j] = (nv > 0.0) ? (float)(1.0 / nv) : 0.0f;
    }
}

// finalize 2: rank by counting (descending, ties by index, NaN first) -> perm[rank] = j
__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ sig, int n_pad, int* __restrict__ perm) {
    const int b = blockIdx.y;
    const float* s = sig + (int64_t)b * n_pad;
    const int j = blockIdx.x * 256 + threadIdx.x;
    __shared__ float buf[256];
    float me = (j < n_pad) ? s[j] : 0.0f;
    if (me != me) me = INFINITY;
    int cnt = 0;
    for (int base = 0; base < n_pad; base += 256) {
        float v = (base + threadIdx.x < n_pad) ? s[base + threadIdx.x] : -INFINITY;
        if (v != v) v = INFINITY;
        buf[threadIdx.x] = v;
        __syncthreads();
        const int lim = min(256, n_pad - base);
        for (int i = 0; i < lim; ++i) {
            const float o = buf[i];
            cnt += (o > me || (o == me && base + i < j)) ? 1 : 0;
        }
        __syncthreads();
    }
    if (j < n_pad) perm[(int64_t)b * n_pad + cnt] = j;
}

// finalize 3: gather the leading k columns, normalised, into row-major outputs.
//   part A rows [0, rowsA)  -> outA [rowsA, k]   (normalised a_j: left vectors of the oriented matrix)
//   part V rows [0, rowsV)  -> outV [rowsV, k]   (normalised v_j: right vectors)
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ X, int64_t panel_stride, int m_pad, int R,
                                                     const float* __restrict__ sig, const float* __restrict__ inv_na,
                                                     const float* __restrict__ inv_nv, const int* __restrict__ perm,
                                                     int rowsA, int rowsV, int k, float* __restrict__ outA,
                                                     float* __restrict__ outV, float* __restrict__ outS) {
    const int jj = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;  // 4 row lanes
    if (jj >= k) return;
    const int j = perm[jj];
    const float* P = X + (int64_t)(j >> 5) * panel_stride + (j & 31);
    const int rtot = rowsA + rowsV;
    const int rb = blockIdx.y * 64;
    if (blockIdx.y == 0 && rl == 0 && outS) outS[jj] = sig[j];
    const float ia = inv_na[j], iv = inv_nv[j];
    for (int r = rb + rl; r < min(rb + 64, rtot); r += 4) {
        if (r < rowsA) {
            if (outA) outA[(int64_t)r * k + jj] = P[(int64_t)r * PB] * ia;
        } else {
            const int i = r - rowsA;
            if (outV) outV[(int64_t)i * k + jj] = P[(int64_t)(m_pad + i) * PB] * iv;
        }
    }
}

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
// unit-diagonal scaling of the upper triangle; empty columns (d = 0) become unit vectors so that the factorisation proceeds
__global__ void chol_scale_kernel(double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int n_pad, const double* __restrict__ d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= n_pad || j < i) return;
    const double di = d[(int64_t)b * n_pad + i], dj = d[(int64_t)b * n_pad + j];
    double* g = G + (int64_t)b * g_batch_stride + (int64_t)i * ldg + j;
    if (i == j) *g = 1.0;
    else *g = (di > 0.0 && dj > 0.0) ? *g / (di * dj) : 0.0;
}

// column pivoting "light": order the columns by decreasing norm before the factorisation (symmetric permutation of the Gram
// matrix).  With sorted columns R is graded and Jacobi on R^T starts much closer to diagonal (Drmac-Veselic preconditioning;
// the CPU prototype needs 9 instead of 12 sweeps at n = 2048).
__global__ void iota_kernel(int* __restrict__ perm, int n_pad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (j < n_pad) perm[(int64_t)b * n_pad + j] = j;
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
// out[perm[i]][:] = in[i][:]   (rows of the right vectors back to the original column order)
__global__ void row_unpermute_kernel(const float* __restrict__ in, const int* __restrict__ perm, int rows, int k, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (c >= k || i >= rows) return;
    out[(int64_t)perm[i] * k + c] = in[(int64_t)i * k + c];
}

constexpr int CB = 64;        // Cholesky block size
constexpr int CLD = CB + 1;   // LDS leading dimension (doubles)

// Block step jb of the right-looking upper Cholesky  G = R^T R  (in place, fp64).  Every workgroup factors the 64x64 diagonal
// block in LDS (redundantly: ~64 short steps) and inverts it; workgroup 0 stores R_jj, workgroup q >= 1 forms the block
// R_{jb, jb+q} = R_jj^-T G_{jb, jb+q}.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, int jb,
                                                         int* __restrict__ fail, double* __restrict__ Dg, int nbk) {
    __shared__ double A[CB * CLD];
    __shared__ double Ri[CB * CLD];
    double* Bs = A;  // the unfactored / factored diagonal block is dead once its inverse exists
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nq = nbk - jb;  // column blocks jb .. nbk-1 of this block row; workgroup w handles q = w, w + gridDim.x, ...
    double* Gb = G + (int64_t)b * g_batch_stride;
    const int64_t o = (int64_t)jb * CB;
    for (int e = tid; e < CB * CB; e += 256) {
        const int i = e >> 6, j = e & 63;
        A[i * CLD + j] = (j >= i) ? Gb[(o + i) * ldg + o + j] : 0.0;
        Ri[i * CLD + j] = 0.0;
    }
    __syncthreads();
    bool bad = false;
    for (int c = 0; c < CB; ++c) {
        double piv = A[c * CLD + c];
        if (!(piv > 1e-13)) { bad = true; piv = 1e-13; }  // unit-diagonal scaling: pivots live in (0, 1]
        const double r = sqrt(piv), rinv = 1.0 / r;
        __syncthreads();
        if (tid == 0) A[c * CLD + c] = r;
        for (int k = c + 1 + tid; k < CB; k += 256) A[c * CLD + k] *= rinv;
        __syncthreads();
        // trailing update of the upper triangle: A[i][k] -= R[c][i] R[c][k], c < i <= k  (16 x 16 thread grid striding the block: no
        // integer division per element, which cost more than the fp64 FMA it addressed)
        for (int i = c + 1 + (tid >> 4); i < CB; i += 16) {
            const double ri = A[c * CLD + i];
            for (int k = c + 1 + (tid & 15); k < CB; k += 16)
                if (k >= i) A[i * CLD + k] -= ri * A[c * CLD + k];
        }
        __syncthreads();
    }
    if (bad && tid == 0) atomicMax(&fail[b], jb + 1);
    // inverse of the upper triangular R_jj: thread j solves R z = e_j by back substitution
    if (tid < CB) {
        const int j = tid;
        Ri[j * CLD + j] = 1.0 / A[j * CLD + j];
        for (int i = j - 1; i >= 0; --i) {
            double acc = 0.0;
            for (int k = i + 1; k <= j; ++k) acc += A[i * CLD + k] * Ri[k * CLD + j];
            Ri[i * CLD + j] = -acc / A[i * CLD + i];
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        // R_jj goes to a side buffer: the other workgroups of this step may still be loading the unfactored A_jj from G
        double* dgo = Dg + ((int64_t)b * nbk + jb) * (CB * CB);
        for (int e = tid; e < CB * CB; e += 256) {
            const int i = e >> 6, j = e & 63;
            dgo[e] = (j >= i) ? A[i * CLD + j] : 0.0;
        }
    }
    for (int q = (blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x); q < nq; q += gridDim.x) {
        __syncthreads();  // previous block's reads of Bs (and the R_jj store above) are done
        const int64_t oc = (int64_t)(jb + q) * CB;
        for (int e = tid; e < CB * CB; e += 256) {
            const int i = e >> 6, j = e & 63;
            Bs[i * CLD + j] = Gb[(o + i) * ldg + oc + j];
        }
        __syncthreads();
        // Y = Ri^T B :  Y[i][c] = sum_{k <= i} Ri[k][i] B[k][c]
        for (int e = tid; e < CB * CB; e += 256) {
            const int i = e >> 6, c = e & 63;
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc += Ri[k * CLD + i] * Bs[k * CLD + c];
            Gb[(o + i) * ldg + oc + c] = acc;
        }
    }
}

// The same block step as two launches: the diagonal factorisation WAVE-LOCAL, one wave per problem, and the block row by fp64 MFMA.
// chol_panel_kernel spends ~150 of its 183 us in 64 column steps of three workgroup barriers each, in every workgroup of the block row
// (the Cholesky of a 4096-column Gram matrix is a chain of 64 such launches: 11.7 ms per step of 32 problems).  chol_diag_wave_kernel holds
// the block with lane = column, registers = rows (64 doubles): a column step is two v_readlane for the pivot, two per row multiplier and one
// v_fma_f64 per remaining row, no barrier; the inverse is a back substitution per lane against R read as LDS broadcasts.  R_jj goes to the
// side buffer Dg (r_to_f32 reads the diagonal blocks there), R_jj^-1 over the block itself, where chol_trsm_kernel — one workgroup per
// block of the block row, Y = R_jj^-T G_jq with v_mfma_f64_16x16x4, wave w: rows 16 w .. 16 w + 15 — reads it.
// (Factorising inside every workgroup of the block row, as chol_panel_kernel does, was measured at 384 us per launch with this wave-local
// form: 2048 one-wave-busy workgroups of 256 VGPRs and 67 KB LDS run in four rounds.)
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

// R (fp32, dense [n_pad][n_pad], zero below the diagonal) = Rs * diag(d);  transposed != 0 writes R^T instead
__global__ void r_to_f32_kernel(const double* __restrict__ G, int64_t ldg, int64_t g_batch_stride, const double* __restrict__ Dg,
                                const double* __restrict__ d, int n_pad, int transposed, float* __restrict__ R, int64_t r_batch_stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= n_pad) return;
    double rv = 0.0;
    if (j >= i) {
        if ((i / CB) == (j / CB)) rv = Dg[((int64_t)b * (n_pad / CB) + i / CB) * (CB * CB) + (i % CB) * CB + (j % CB)];  // diagonal blocks
        else rv = G[(int64_t)b * g_batch_stride + (int64_t)i * ldg + j];
    }
    const double v = rv * d[(int64_t)b * n_pad + j];
    float* Rb = R + (int64_t)b * r_batch_stride;
    if (!transposed) Rb[(int64_t)i * n_pad + j] = (float)v;
    else Rb[(int64_t)j * n_pad + i] = (float)v;
}

// R^T through 32x32 LDS tiles: r_to_f32_kernel with transposed != 0 stores one float per 4-byte-strided row, a column at a time
// (3.4 ms per 32 x 4096^2); here both the fp64 reads and the fp32 stores run along rows.  grid (n_pad/32, n_pad/32, batch), 256 threads.
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
// Workgroup tile 128 x 128; wave w owns rows 32w..32w+31; per panel: A tile through a private padded LDS image (row-per-lane
// b128 reads), B tile (32 x 128 of Vr) shared by the 4 waves through LDS.
constexpr int NBLD = 132;  // LDS leading dimension of the B tile (floats)
__global__ __launch_bounds__(256) void nn_gemm_kernel(const float* __restrict__ X, int64_t panel_stride, int nb, int rows, int cols,
                                                      const float* __restrict__ Vr, int64_t ldv, const float* __restrict__ S, int k,
                                                      float* __restrict__ out, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float At[4][32 * 36];
    __shared__ __attribute__((aligned(16))) float Bt[32 * NBLD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
    const int h = lane >> 5, c = lane & 31;
    const int r0 = blockIdx.y * 128 + w * 32;
    const int c0 = blockIdx.x * 128;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x16){0};
    float* my = At[w];
    // the next panel's A rows (4 x 16 B per lane) and B rows (16 floats per thread) are prefetched into registers while the
    // current panel is in the matrix pipe
    f32x4 pa[4];
    float pb[16];
    auto fetch = [&](int p) {
        const float* P = X + (int64_t)p * panel_stride;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + lane * 4;
            const int row = idx >> 5, col = idx & 31;
            pa[it] = (r0 + row < rows) ? *(const f32x4*)(P + (int64_t)(r0 + row) * PB + col) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 256 + tid;
            const int kr = idx >> 7, cc = idx & 127;
            const int vr = p * PB + kr, vc = c0 + cc;
            pb[it] = (vr < cols && vc < k) ? Vr[(int64_t)vr * ldv + vc] : 0.0f;
        }
    };
    fetch(0);
    for (int p = 0; p < nb; ++p) {
        __syncthreads();  // previous panel's Bt fully consumed
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + lane * 4;
            *(f32x4*)(my + (idx >> 5) * 36 + (idx & 31)) = pa[it];
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 256 + tid;
            Bt[(idx >> 7) * NBLD + (idx & 127)] = pb[it];
        }
        __syncthreads();
        if (p + 1 < nb) fetch(p + 1);
        float a[16];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const f32x4 v = *(const f32x4*)(my + c * 36 + h * 16 + t4 * 4);
            a[4 * t4 + 0] = v[0]; a[4 * t4 + 1] = v[1]; a[4 * t4 + 2] = v[2]; a[4 * t4 + 3] = v[3];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], Bt[(h * 16 + t) * NBLD + tl * 32 + c], acc[tl], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const int col = c0 + tl * 32 + c;
        if (col >= k) continue;
        const float sv = S ? S[col] : 1.0f;
        const float inv = sv > 0.0f ? 1.0f / sv : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < rows) out[(int64_t)row * ldo + col] = acc[tl][reg] * inv;
        }
    }
}

// Split-bf16 form of nn_gemm_kernel (twolevel.h arithmetic: every fp32 operand = three bf16 exactly, six products per fp32 product on
// the bf16 matrix pipe, fp32 accumulation): the long-side GEMM was 62 % of the fp32-MFMA peak and 22 ms per 16 problems.  The
// fetching thread splits its 8 consecutive k-values once and stores them as ready MFMA operands ([block][k-step][part][lane] images,
// 36-operand half blocks so the scattered A writes fall on distinct banks); a wave then issues 48 bf16 MFMAs per 32-column panel
// against 30 ds_read_b128, no VALU in the inner loop.  Same tiling (128 x 128 per workgroup, wave w = rows 32 w ..), same epilogue.
constexpr int NG_HB = 36, NG_BLK = 2 * NG_HB;
__global__ __launch_bounds__(256, 2) void nn_gemm_split_kernel(const float* __restrict__ X, int64_t panel_stride, int nb, int rows, int cols,
                                                               const float* __restrict__ Vr, int64_t ldv, const float* __restrict__ S, int k,
                                                               float* __restrict__ out, int64_t ldo) {
    __shared__ u32x4 Aimg[4 * 2 * 3 * NG_BLK];
    __shared__ u32x4 Bimg[4 * 2 * 3 * NG_BLK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
    const int h = lane >> 5, c = lane & 31;
    const int r0 = blockIdx.y * 128;
    const int c0 = blockIdx.x * 128;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x16){0};
    // A pieces: q = tid + 256 j -> row q >> 2 of the 128, k-chunk q & 3 (8 values = 32 B); B pieces: column tid & 127, k-chunk (tid >> 7) + 2 j
    f32x4 pa[2][2];
    float pb[2][8];
    int adst[2], bdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + 256 * j, row = q >> 2, kc = q & 3;
        adst[j] = (((row >> 5) * 2 + (kc >> 1)) * 3) * NG_BLK + (kc & 1) * NG_HB + (row & 31);
        const int cc = tid & 127, kb = (tid >> 7) + 2 * j;
        bdst[j] = (((cc >> 5) * 2 + (kb >> 1)) * 3) * NG_BLK + (kb & 1) * NG_HB + (cc & 31);
    }
    auto fetch = [&](int p) {
        const float* P = X + (int64_t)p * panel_stride;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + 256 * j, row = q >> 2, kc = q & 3;
            if (r0 + row < rows) {
                const float* src = P + (int64_t)(r0 + row) * PB + 8 * kc;
                pa[j][0] = *(const f32x4*)src;
                pa[j][1] = *(const f32x4*)(src + 4);
            } else {
                pa[j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                pa[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int cc = tid & 127, kb = (tid >> 7) + 2 * j, vc = c0 + cc;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int vr = p * PB + 8 * kb + e;
                pb[j][e] = (vr < cols && vc < k) ? Vr[(int64_t)vr * ldv + vc] : 0.0f;
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
    fetch(0);
    for (int p = 0; p < nb; ++p) {
        __syncthreads();  // previous panel's images fully consumed
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            put(Aimg + adst[j], pa[j][0][0], pa[j][0][1], pa[j][0][2], pa[j][0][3], pa[j][1][0], pa[j][1][1], pa[j][1][2], pa[j][1][3]);
            put(Bimg + bdst[j], pb[j][0], pb[j][1], pb[j][2], pb[j][3], pb[j][4], pb[j][5], pb[j][6], pb[j][7]);
        }
        __syncthreads();
        if (p + 1 < nb) fetch(p + 1);
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
    }
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
__global__ __launch_bounds__(256) void colsumsq_kernel(const float* __restrict__ Y, int64_t ldy, int rows, int k, int rows_per_split,
                                                       double* __restrict__ part) {
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
__global__ __launch_bounds__(256) void colfinish_kernel(const double* __restrict__ part, int nsplit, int k, float* __restrict__ S,
                                                        float* __restrict__ inv) {
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
__global__ void colscale_kernel(float* __restrict__ Y, int64_t ldy, int rows, int k, const float* __restrict__ inv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r0 = blockIdx.y * 32;
    if (c >= k) return;
    const float f = inv[c];
    for (int r = r0; r < min(r0 + 32, rows); ++r) Y[(int64_t)r * ldy + c] *= f;
}

// --------------------------------------------------------------------------------------------------
// stream groups of the sweeps: 1 unless ASVD_GROUPS asks for more (see common.h "cache maintenance": several groups need fences that
// cost more than the overlap gains)
static int stream_groups_for(int batch) {
    const char* e = getenv("ASVD_GROUPS");
    int g = e ? atoi(e) : 1;
    if (g < 1) g = 1;
    if (g > 4) g = 4;
    return g > batch ? batch : g;
}

static bool pair_order_xor() {
    const char* e = getenv("ASVD_ORDER");
    return !(e && !strncmp(e, "rr", 2));
}

struct Plan {
    int batch;
    int64_t m, n;         // as given
    int transposed;       // oriented = A^T when m < n
    int rows, cols;       // oriented dims, rows >= cols
    int m_pad, n_pad, nb, npairs, R, R_upd, want_v, vmode;  // vmode: 0 none, 1 accumulate V in the sweeps, 2 backsolve at the end
    int nsplit, rows_per_split, rows_per_wg, nchunks;
    int fused, nchunks_f, rows_per_wg_f;  // upgram path (XOR ordering, power-of-two panel count)
    // two-level dense sweeps (twolevel.h): ns super-panels of 64 columns, npairs_s pair slots per super-step (power-of-two padded)
    int two, ns, npairs_s, nsplit_s, rows_per_split_s, nchunks_s, rows_per_wg_s, nchunks_q, rows_per_wg_q;
    size_t off_gd32, off_gx6, off_q0, off_d0, off_qfin, off_subact;
    int64_t panel_stride, batch_stride;
    // workspace offsets in bytes
    size_t off_x, off_xorig, off_gpart, off_q, off_active, off_sig, off_ina, off_inv, off_perm, off_flags, off_pflag, off_plist, total;
};

int make_plan(int batch, int64_t m, int64_t n, int want_u, int want_vv, Plan& p) {
    if (batch < 1 || m < 1 || n < 1 || m > (1 << 24) || n > (1 << 24)) return ASVD_E_BADARG;
    p.batch = batch;
    p.m = m;
    p.n = n;
    p.transposed = (m < n) ? 1 : 0;
    p.rows = (int)(p.transposed ? n : m);
    p.cols = (int)(p.transposed ? m : n);
    p.m_pad = (int)round_up64(p.rows, 32);
    p.n_pad = (int)round_up64(p.cols, PW);
    p.nb = p.n_pad / PB;
    {
        int pw2 = 2;
        while (pw2 < p.nb) pw2 <<= 1;
        p.npairs = pair_order_xor() ? pw2 / 2 : p.nb / 2;  // XOR ordering runs over the panel count padded to a power of two
    }
    // want_u / want_vv: left / right vectors OF THE ORIENTED problem (columns of the rotated matrix / backsolved V rows)
    p.want_v = (want_u || want_vv) ? 1 : 0;
    p.vmode = 0;
    if (want_vv) {
        const char* e = getenv("ASVD_VMODE");
        p.vmode = (e && strcmp(e, "accumulate") == 0) ? 1 : 2;
    }
    p.R = p.m_pad + (want_vv ? p.n_pad : 0);
    p.R_upd = (p.vmode == 1) ? p.R : p.m_pad;
    // gram: choose the row split that minimises (rounds of resident workgroups) x (chunks per wave + fixed overhead).
    // 3 workgroups of 4 waves fit per CU (148 VGPR+AGPR) -> 768 slots; a grid of 1.3 x slots costs 2 full rounds.
    {
        const int launch_batch = (int)ceil_div64(batch, stream_groups_for(batch));  // problems per launch (stream groups)
        const int64_t nchunk_total = p.m_pad / 32;
        int64_t best_ns = 1;
        double best_cost = 1e300;
        for (int64_t ns = 1; ns <= nchunk_total && ns <= 64; ++ns) {
            const int64_t wgs = ns * p.npairs * launch_batch;
            const int64_t rounds = ceil_div64(wgs, 768);
            const int64_t chunks_wg = ceil_div64(nchunk_total, ns);
            const int64_t chunks_wave = ceil_div64(chunks_wg, 4);
            const double cost = (double)rounds * ((double)chunks_wave + 1.5) + 0.02 * ns;  // mild penalty: partials traffic
            if (cost < best_cost) { best_cost = cost; best_ns = ns; }
        }
        p.rows_per_split = (int)(ceil_div64(nchunk_total, best_ns) * 32);
        p.nsplit = (int)ceil_div64(p.m_pad, p.rows_per_split);
    }
    // update: 128-row iterations; aim for >= 1024 workgroups but >= 2 iterations per workgroup when possible
    int64_t wantc = ceil_div64(1024, (int64_t)p.npairs * ceil_div64(batch, stream_groups_for(batch)));  // per launch = one stream group
    int64_t iters_total = ceil_div64(p.R_upd, 128);
    int64_t nc = wantc < 1 ? 1 : (wantc > iters_total ? iters_total : wantc);
    p.rows_per_wg = (int)(ceil_div64(iters_total, nc) * 128);
    p.nchunks = (int)ceil_div64(p.R_upd, p.rows_per_wg);
    {
        // upgram: one workgroup per CU; aim for ~4 workgroups per CU-slot over the launch, >= 128 rows per workgroup
        const char* ef = getenv("ASVD_FUSED");
        // opt-in (ASVD_FUSED=1): measured +4 % on the 16 x 4096^2 bench but -3 % on the mixed-shape full-model run — at one
        // workgroup per CU the kernel is latency-sensitive; kept as the building block for the split-precision / two-level plan
        p.fused = (pair_order_xor() && p.nb == 2 * p.npairs && p.nb >= 4 && ef && atoi(ef) == 1) ? 1 : 0;
        const int launch_batch = (int)ceil_div64(batch, stream_groups_for(batch));
        const int64_t quads = (int64_t)(p.nb / 4) * launch_batch;
        int64_t want = ceil_div64(1024, quads > 0 ? quads : 1);
        const int64_t iters = ceil_div64(p.R_upd, 128);
        if (want < 1) want = 1;
        if (want > iters) want = iters;
        p.rows_per_wg_f = (int)(ceil_div64(iters, want) * 128);
        p.nchunks_f = (int)ceil_div64(p.R_upd, p.rows_per_wg_f);
    }
    {
        // two-level dense sweeps: default for >= 8 panels under the XOR ordering (ASVD_TWOLEVEL=0 restores the single-level sweep)
        const char* e2 = getenv("ASVD_TWOLEVEL");
        p.two = (pair_order_xor() && p.nb >= 8 && !p.fused && !(e2 && atoi(e2) == 0)) ? 1 : 0;
        p.ns = p.nb / 2;
        int pw2 = 2;
        while (pw2 < p.ns) pw2 <<= 1;
        p.npairs_s = pw2 / 2;
        const int launch_batch = stream_groups_for(batch) > 1 ? (int)ceil_div64(batch, stream_groups_for(batch)) : (batch + 1) / 2;  // one stream: two pipelined halves
        // sgram6: 3 workgroups (32 KiB LDS, ~150 VGPRs) per CU -> 768 slots; 16-row chunks per wave, same cost model as the single-level Gram
        const int64_t nchunk_total = p.m_pad / 32;
        int64_t best_ns = 1;
        double best_cost = 1e300;
        for (int64_t ns = 1; ns <= nchunk_total && ns <= 64; ++ns) {
            const int64_t wgs = ns * p.npairs_s * launch_batch;
            const int64_t rounds = ceil_div64(wgs, 768);
            const int64_t chunks_wave = ceil_div64(2 * ceil_div64(nchunk_total, ns), 4);
            const double cost = (double)rounds * ((double)chunks_wave + 1.5) + 0.02 * ns;
            if (cost < best_cost) { best_cost = cost; best_ns = ns; }
        }
        p.rows_per_split_s = (int)(ceil_div64(nchunk_total, best_ns) * 32);
        p.nsplit_s = (int)ceil_div64(p.m_pad, p.rows_per_split_s);
        // supdate: 32-row tiles; aim for >= 1024 workgroups per launch and >= 4 tiles per workgroup
        const int64_t tiles = ceil_div64(p.R_upd, 32);
        int64_t wantc = ceil_div64(1024, (int64_t)p.npairs_s * launch_batch);
        int64_t nc = std::max<int64_t>(1, std::min<int64_t>(wantc, ceil_div64(tiles, 4)));
        p.rows_per_wg_s = (int)(ceil_div64(tiles, nc) * 32);
        p.nchunks_s = (int)ceil_div64(p.R_upd, p.rows_per_wg_s);
        // supgram (update of a step fused with the Gram tiles of the next): one 512-thread workgroup per CU and quad of four
        // super-panels; ~512 workgroups per launch where the rows allow >= 8 tiles each
        // measured (16 x 4096^2): 2 chunks (512 workgroups) beat 4 and 8 — longer streams per workgroup, fewer partial tiles for the solves to sum
        const int64_t quads = std::max<int64_t>(1, pw2 / 4) * batch;
        // (and 1 chunk = exactly one workgroup per CU beats 2 once the quads alone fill the chip: 170.7 vs 174.0 ms per step, solves 97 vs 100)
        // small batches: one round of 256 workgroups too (batch 4: 43.7 ms of supgram per step with 4 chunks, 47.3 with 8, 52.8 with 16);
        // the solves sum the partial tiles with independent loads, so their cost no longer grows with the chunk count
        int64_t nq = std::max<int64_t>(1, std::min<int64_t>(ceil_div64(256, quads), std::max<int64_t>(1, tiles / 8)));
        {
            // counts that are not a power of two (13B: 80 super-panels = 20 real quads per problem in a 32-quad grid): one workgroup per CU means
            // the launch runs in whole rounds of 256 workgroups, and 320 real workgroups cost two rounds (measured 1179 us per launch at
            // 16 x 5120^2; four row chunks = exactly five rounds of a quarter length: 737 us).  Pick the chunk count that wastes the least.
            const int64_t real = ceil_div64(p.ns, 4) * batch;
            if (real != quads && real >= 256) {
                double best = 1e300;
                for (int64_t c = 1; c <= 8 && c <= std::max<int64_t>(1, tiles / 8); c *= 2) {
                    const double cost = (double)ceil_div64(real * c, 256) / (double)c * (1.0 + 0.03 * (double)c);
                    if (cost < best) { best = cost; nq = c; }
                }
            }
        }
        if (getenv("ASVD_SUPGRAM_CHUNKS")) nq = std::max<int64_t>(1, std::min<int64_t>(atoi(getenv("ASVD_SUPGRAM_CHUNKS")), tiles));
        p.rows_per_wg_q = (int)(ceil_div64(tiles, nq) * 32);
        p.nchunks_q = (int)ceil_div64(p.R_upd, p.rows_per_wg_q);
    }
    p.panel_stride = (int64_t)p.R * PB;
    p.batch_stride = p.panel_stride * p.nb;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    p.off_x = take((size_t)p.batch_stride * batch * sizeof(float));
    p.off_xorig = take(p.vmode == 2 ? (size_t)p.m_pad * PB * p.nb * batch * sizeof(float) : 0);
    p.off_gpart = take((size_t)batch * p.npairs * std::max(p.nsplit, p.fused ? p.nchunks_f : 0) * 3072 * sizeof(float));
    p.off_q = take((size_t)batch * p.npairs * PW * PW * sizeof(float));
    p.off_active = take((size_t)batch * p.npairs * sizeof(int));
    p.off_sig = take((size_t)batch * p.n_pad * sizeof(float));
    p.off_ina = take((size_t)batch * p.n_pad * sizeof(float));
    p.off_inv = take((size_t)batch * p.n_pad * sizeof(float));
    p.off_perm = take((size_t)batch * p.n_pad * sizeof(int));
    p.off_flags = take((size_t)batch * 4 * sizeof(int));  // [maxoff bits | nrot | done | super-pair updates] x batch (SoA)
    p.off_pflag = take((size_t)batch * p.nb * p.nb);      // sparse-sweep pair marks
    p.off_plist = take((size_t)batch * p.nb * p.nb * sizeof(int));  // per-step lists of marked pairs (bound: steps x nb/2 slots per problem)
    const size_t t2 = p.two ? 1 : 0;
    p.off_gd32 = take(t2 * batch * p.nb * 1024 * sizeof(float));                          // carried 32x32 diagonal blocks, one per panel
    p.off_gx6 = take(t2 * batch * p.npairs_s * std::max(p.nsplit_s, p.nchunks_q) * 6 * 1024 * sizeof(float));  // sgram6 / supgram partial tiles
    p.off_q0 = take(t2 * batch * p.npairs_s * 2 * PW * PW * sizeof(float));               // Q of the two step-0 solves
    p.off_d0 = take(t2 * batch * p.npairs_s * 4 * 1024 * sizeof(float));                  // diagonal blocks after step 0
    p.off_qfin = take(t2 * batch * p.npairs_s * SP * SP * sizeof(float));                 // Q^(0) Q^(1) of every super-pair
    p.off_subact = take(t2 * batch * p.npairs_s * 4 * sizeof(int));
    p.total = off;
    return ASVD_OK;
}

// super-panel pairs by round-robin tournament instead of a padded XOR schedule: only where the super-panel count is not a power of two
static bool super_rr_for(const Plan& p) {
    if (!p.two || (p.ns & (p.ns - 1)) == 0) return false;
    const char* e = getenv("ASVD_SUPER_RR");
    return e && atoi(e) == 1;
}
// group pairs of every round of the grouped schedule: circle method over the ns / 16 groups (+ a bye when their number is odd)
static void set_group_table(Sched& sc, int ns) {
    const int ng = ns / 16, n = ng + (ng & 1), gm = ng / 2;
    signed char (*tab)[4][2] = sc.gpair;
    std::memset(sc.gpair, 0, sizeof(sc.gpair));
    for (int r = 0; r < n - 1; ++r) {
        int m = 0;
        for (int k = 0; k < n / 2; ++k) {
            const int a = (k == 0) ? 0 : 1 + (k - 1 + r) % (n - 1);
            const int pb = n - 1 - k;
            const int bb = 1 + (pb - 1 + r) % (n - 1);
            if (a >= ng || bb >= ng) continue;  // the bye
            tab[r][m][0] = (signed char)std::min(a, bb);
            tab[r][m][1] = (signed char)std::max(a, bb);
            ++m;
        }
    }
    sc.gm = gm;
}

// grouped schedule (super_pair, c_super_order = 2): ns a multiple of 16, not a power of two, at most 8 groups
static bool grouped_applies(int ns) {  // a multiple of 16, not a power of two, at most 8 groups, and fewer super-steps than the padded XOR schedule
    if ((ns & (ns - 1)) == 0 || (ns % 16) || ns / 16 > 8) return false;
    int pw2 = 2;
    while (pw2 < ns) pw2 <<= 1;
    const int ng = ns / 16;
    return 15 + 16 * ((ng & 1) ? ng : ng - 1) < pw2 - 1;
}
static bool super_grouped_for(const Plan& p) {
    if (!p.two || !grouped_applies(p.ns) || super_rr_for(p)) return false;
    const char* e = getenv("ASVD_SUPER_GROUPED");  // default on; =0 restores the padded XOR schedule
    return !(e && atoi(e) == 0);
}
static int super_grouped_rounds(const Plan& p) { const int ng = p.ns / 16; return (ng & 1) ? ng : ng - 1; }

// ---- optional per-class timing with HIP events on the call's stream ------------------------------
// profiling state is per host thread: concurrent calls from different threads (on their own streams and workspaces) do not share it
thread_local bool g_prof_enabled = false;
// classes: 0 pack / reduce, 1 two-level Gram pass, 2 eigen-solves, 3 two-level update pass, 4 finalize, 5 coupling snapshot,
//          6 single-level Gram, 7 single-level update
constexpr int NPROF = 9;
thread_local float g_prof_ms[NPROF] = {0};
thread_local int g_prof_launches[NPROF] = {0};
thread_local long long g_prof_pairs[3] = {0, 0, 0};  // 32-panel pair visits, rotated 32-panel pairs, updated super-pairs (two-level sweeps)
thread_local std::vector<float> g_prof_sweep_ms;       // wall time of every sweep of the last profiled call (all problems of the batch together)
thread_local std::vector<long long> g_prof_sweep_rot;  // pairs rotated in it  // {pair visits (gram), rotated pairs (evd + update)} of the last profiled call
struct ProfRec { int cls; hipEvent_t a, b; };
thread_local std::vector<ProfRec> g_prof_recs;

struct ProfScope {
    int cls; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(int c, hipStream_t s) : cls(c), st(s), on(g_prof_enabled) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(b, st); g_prof_recs.push_back({cls, a, b}); }
    }
};
void prof_begin() {
    for (int i = 0; i < NPROF; ++i) { g_prof_ms[i] = 0; g_prof_launches[i] = 0; }
    g_prof_pairs[0] = g_prof_pairs[1] = g_prof_pairs[2] = 0;
    g_prof_sweep_ms.clear();
    g_prof_sweep_rot.clear();
    g_prof_recs.clear();
}
void prof_end() {
    for (auto& r : g_prof_recs) {
        float ms = 0;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        g_prof_ms[r.cls] += ms;
        g_prof_launches[r.cls] += 1;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_recs.clear();
}

template <int DT>
int launch_pack(const void* src, int64_t ld, const void* s, int cs_dtype, const Plan& p, float* Xb, hipStream_t st) {
    dim3 grid((unsigned)ceil_div64(p.rows, 32), (unsigned)ceil_div64(p.cols, 256));
    const int has = s ? 1 : 0;
    switch (cs_dtype) {
        case ASVD_F32: pack_kernel<DT, ASVD_F32><<<grid, 256, 0, st>>>(src, ld, s, has, p.transposed, p.rows, p.cols, p.R, Xb); break;
        case ASVD_F16: pack_kernel<DT, ASVD_F16><<<grid, 256, 0, st>>>(src, ld, s, has, p.transposed, p.rows, p.cols, p.R, Xb); break;
        case ASVD_BF16: pack_kernel<DT, ASVD_BF16><<<grid, 256, 0, st>>>(src, ld, s, has, p.transposed, p.rows, p.cols, p.R, Xb); break;
        default: return ASVD_E_BADARG;
    }
    return ASVD_OK;
}

}  // namespace

extern "C" {

void asvd_svd_set_profiling(int enabled) { g_prof_enabled = enabled != 0; }
int asvd_svd_get_sweep_times(float* ms_host, long long* rotated_host, int cap) {
    const int n = (int)g_prof_sweep_ms.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (ms_host) ms_host[i] = g_prof_sweep_ms[i];
        if (rotated_host) rotated_host[i] = g_prof_sweep_rot[i];
    }
    return n;
}

int asvd_svd_get_pair_counts(long long* counts_host) {
    if (!counts_host) return ASVD_E_BADARG;
    counts_host[0] = g_prof_pairs[0];
    counts_host[1] = g_prof_pairs[1];
    counts_host[2] = g_prof_pairs[2];
    return ASVD_OK;
}

int asvd_svd_get_profile(float* ms_host, int* launches_host) {
    if (!ms_host || !launches_host) return ASVD_E_BADARG;
    for (int i = 0; i < NPROF; ++i) { ms_host[i] = g_prof_ms[i]; launches_host[i] = g_prof_launches[i]; }
    return ASVD_OK;
}

// forward declarations of the tall-path helpers (defined after the direct driver)
static bool tall_wanted(const Plan& p);
static size_t tall_worksize(int batch, int64_t m, int64_t n, int want_vectors, int64_t k);

int asvd_svd_worksize(int batch, int64_t m, int64_t n, int want_vectors, size_t* bytes) {
    if (!bytes) return ASVD_E_BADARG;
    Plan p;
    int rc = make_plan(batch, m, n, want_vectors, want_vectors, p);
    if (rc) return rc;
    size_t need = p.total;  // the direct path is always available as the fallback
    if (tall_wanted(p)) {
        const size_t t = tall_worksize(batch, m, n, want_vectors, p.cols);
        if (t > need) need = t;
    }
    *bytes = need;
    return ASVD_OK;
}

static int svd_direct(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                      const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                      float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                      int* info_host, void* stream, bool manage_profile) {
    if (!a_host || !S_host || !work || !dtype_ok(a_dtype) || lda < n) return ASVD_E_BADARG;
    if (cs_host && !dtype_ok(cs_dtype)) return ASVD_E_BADARG;
    Plan p;
    {
        // vectors of the oriented problem: its left vectors are U of A (not transposed) or V of A (transposed)
        const bool tr = m < n;
        const int want_left = tr ? (V_host != nullptr) : (U_host != nullptr);
        const int want_right = tr ? (U_host != nullptr) : (V_host != nullptr);
        int rc0 = make_plan(batch, m, n, want_left, want_right, p);
        if (rc0) return rc0;
    }
    if (k < 1 || k > p.cols) return ASVD_E_BADARG;
    if (work_bytes < p.total) return ASVD_E_WORKSPACE;
    for (int b = 0; b < batch; ++b)
        if (!a_host[b] || !S_host[b]) return ASVD_E_BADARG;
    if (max_sweeps <= 0) max_sweeps = 30;
    if (!(tol > 0.0f)) tol = 1e-6f;
    hipStream_t st = (hipStream_t)stream;

    char* wb = (char*)work;
    float* X = (float*)(wb + p.off_x);
    float* Gpart = (float*)(wb + p.off_gpart);
    float* Qbuf = (float*)(wb + p.off_q);
    int* active = (int*)(wb + p.off_active);
    float* sig = (float*)(wb + p.off_sig);
    float* ina = (float*)(wb + p.off_ina);
    float* inv = (float*)(wb + p.off_inv);
    int* perm = (int*)(wb + p.off_perm);
    unsigned* maxoff = (unsigned*)(wb + p.off_flags);
    int* nrot = (int*)(wb + p.off_flags) + batch;
    int* nupd = (int*)(wb + p.off_flags) + 3 * batch;  // instrumentation: super-pair updates of the sweep
    int* done = (int*)(wb + p.off_flags) + 2 * batch;

    if (g_prof_enabled && manage_profile) prof_begin();

    // ---- pack ----
    {
        ProfScope ps(0, st);
        ASVD_HIP_CHECK(hipMemsetAsync(X, 0, (size_t)p.batch_stride * batch * sizeof(float), st));
        ASVD_HIP_CHECK(hipMemsetAsync(wb + p.off_flags, 0, (size_t)batch * 4 * sizeof(int), st));
        for (int b = 0; b < batch; ++b) {
            float* Xb = X + (int64_t)b * p.batch_stride;
            const void* s = cs_host ? cs_host[b] : nullptr;
            int prc;
            switch (a_dtype) {
                case ASVD_F32: prc = launch_pack<ASVD_F32>(a_host[b], lda, s, cs_dtype, p, Xb, st); break;
                case ASVD_F16: prc = launch_pack<ASVD_F16>(a_host[b], lda, s, cs_dtype, p, Xb, st); break;
                default: prc = launch_pack<ASVD_BF16>(a_host[b], lda, s, cs_dtype, p, Xb, st); break;
            }
            if (prc) return prc;
            if (p.vmode == 1) vinit_kernel<<<(unsigned)ceil_div64(p.cols, 256), 256, 0, st>>>(Xb, p.cols, p.R, p.m_pad);
        }
        if (p.vmode == 2) {
            // keep the packed original (A rows of every panel) for the final backsolve
            ASVD_HIP_CHECK(hipMemcpy2DAsync(wb + p.off_xorig, (size_t)p.m_pad * PB * sizeof(float), X, (size_t)p.R * PB * sizeof(float),
                                            (size_t)p.m_pad * PB * sizeof(float), (size_t)p.nb * batch, hipMemcpyDeviceToDevice, st));
        }
    }

    // ---- sweeps ----
    std::vector<int> flags((size_t)batch * 4, 0);
    std::vector<int> sweeps_done(batch, 0), last_rot(batch, 0), status(batch, ASVD_N_NOCONV);
    std::vector<int> host_done(batch, 0);
    std::vector<float> last_off(batch, 0.0f), prev_off(batch, 1e30f);
    const bool debug = getenv("ASVD_DEBUG") != nullptr;
    // the call's schedule description: a kernel argument of every launch below (no process-global device state)
    Sched sc = default_sched();
    {
        sc.pair_order = pair_order_xor() ? 1 : 0;
        // agent-scope fences at kernel boundaries: on with several stream groups (common.h); ASVD_FENCE=0/1 overrides (experiments)
        sc.fence = stream_groups_for(batch) > 1 ? 3 : 0;
        if (getenv("ASVD_FENCE")) { const int f = atoi(getenv("ASVD_FENCE")); sc.fence = f == 1 ? 3 : (f == 2 ? 1 : (f == 3 ? 2 : (f == 4 ? 4 : (f == 8 ? 8 : (f == 12 ? 12 : 0))))); }
        sc.super_order = super_grouped_for(p) ? 2 : (super_rr_for(p) ? 0 : 1);
        if (sc.super_order == 2) set_group_table(sc, p.ns);
        sc.dbg_fill = getenv("ASVD_EVD_LDSFILL") ? atoi(getenv("ASVD_EVD_LDSFILL")) : 0;
        sc.evd_pairs = getenv("ASVD_EVD_PAIRS") ? std::max(1, std::min(64, atoi(getenv("ASVD_EVD_PAIRS")))) : PW / 2;
    }
    const int nsteps = pair_order_xor() ? 2 * p.npairs - 1 : p.nb - 1;
    // panels whose convergence is enforced: those holding the k leading columns, plus one panel of margin
    const int kb = (int)(ceil_div64(k, PB) + 1 < p.nb ? ceil_div64(k, PB) + 1 : p.nb);
    // one inner sweep of the 64x64 eigen-solve per visit: with the XOR schedule a second one no longer saves outer sweeps
    // (measured 26.3 vs 25.9 SVD/s; under the round-robin order two inner sweeps cut 15 -> 13 outer sweeps)
    const int inner_sweeps = getenv("ASVD_INNER") ? atoi(getenv("ASVD_INNER")) : 1;
    int sweep = 0;
    // Independent problems of a batch are split into two groups driven on two internal streams: while one group sits in its
    // LDS/VALU-bound evd phase the other streams panels through its HBM-bound gram/update phase (different resources).
    constexpr int MAXG = 4;
    int ngroups = stream_groups_for(batch);
    if (ngroups < 1) ngroups = 1;
    if (ngroups > MAXG) ngroups = MAXG;
    if (ngroups > batch) ngroups = batch;
    hipStream_t gst[MAXG] = {st, st, st, st};
    int gb0[MAXG] = {0, 0, 0, 0}, gnb[MAXG] = {batch, 0, 0, 0};
    hipEvent_t ev_fork = nullptr, ev_join[MAXG] = {nullptr, nullptr, nullptr, nullptr};
    if (ngroups >= 2) {
        // side streams belong to the device that is current for this call (one set per device, created on first use)
        constexpr int MAXDEV = 16;
        static hipStream_t s_streams_dev[MAXDEV][MAXG] = {};
        int devid = 0;
        ASVD_HIP_CHECK(hipGetDevice(&devid));
        if (devid < 0 || devid >= MAXDEV) return ASVD_E_BADARG;
        hipStream_t* s_streams = s_streams_dev[devid];
        int off = 0;
        for (int g = 0; g < ngroups; ++g) {
            if (!s_streams[g]) ASVD_HIP_CHECK(hipStreamCreateWithFlags(&s_streams[g], hipStreamNonBlocking));
            gst[g] = s_streams[g];
            gnb[g] = batch / ngroups + (g < batch % ngroups ? 1 : 0);
            gb0[g] = off;
            off += gnb[g];
        }
        ASVD_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        ASVD_HIP_CHECK(hipEventRecord(ev_fork, st));
        for (int g = 0; g < ngroups; ++g) {
            ASVD_HIP_CHECK(hipStreamWaitEvent(gst[g], ev_fork, 0));
            ASVD_HIP_CHECK(hipEventCreateWithFlags(&ev_join[g], hipEventDisableTiming));
        }
    }
    // Step schedule of a dense sweep (XOR distances d, stored as step = d-1).  The local levels d = 1..L are run twice at the start of
    // every sweep (ASVD_DUP=L): the strongest couplings of the sorted, preconditioned matrix sit between neighbouring panels, and a
    // second pass over them is cheap (L extra steps of P-1) — see DESIGN.md 7.
    std::vector<int> sched;
    {
        // default L = min(7, P/16 - 1): measured at 4096^2 (P = 128) 8 -> 7 sweeps, +2.4 %; L = 31 reaches 6 sweeps but loses on steps
        const int P2 = nsteps + 1;
        const int dflt = std::max(0, std::min(7, P2 / 16 - 1));
        const int dup = (pair_order_xor() && !p.fused) ? (getenv("ASVD_DUP") ? atoi(getenv("ASVD_DUP")) : dflt) : 0;
        const int L = std::min(dup, nsteps);
        for (int d = 1; d <= L; ++d) sched.push_back(d - 1);
        for (int st2 = 0; st2 < nsteps; ++st2) sched.push_back(st2);
    }
    // ---- sparse-sweep state (see fullcheck_kernel) ----
    unsigned char* pflag = (unsigned char*)(wb + p.off_pflag);
    int* plist_dev = (int*)(wb + p.off_plist);
    float* dnorm = (float*)(wb + p.off_ina);  // squared column norms of the snapshot (finalize overwrites this buffer later)
    const bool sparse_allowed = pair_order_xor() && p.nb >= 8 && !(getenv("ASVD_SPARSE") && atoi(getenv("ASVD_SPARSE")) == 0);
    const double sparse_frac = getenv("ASVD_SPARSE_FRAC") ? atof(getenv("ASVD_SPARSE_FRAC")) : 0.5;
    bool sparse = false;
    // split-bf16 arithmetic (twolevel.h) for the update pass and the coupling snapshot: on unless ASVD_SPLIT=0.
    // ASVD_PIPE=1 (opt-in, measured slower: 77 vs 69 ms per dense sweep of 16 x 4096^2) rides the eigen-solves of one half of the batch
    // on the Gram launches of the other half (twolevel.h "dual launches").
    const bool split_on = !(getenv("ASVD_SPLIT") && atoi(getenv("ASVD_SPLIT")) == 0);
    const bool split_piped = split_on && getenv("ASVD_PIPE") && atoi(getenv("ASVD_PIPE")) == 1;
    const bool split_check = split_on;
    std::vector<unsigned char> hflag;
    std::vector<int> hlist;
    std::vector<int> sl_off((size_t)MAXG * (nsteps > 0 ? nsteps : 1), 0), sl_cnt((size_t)MAXG * (nsteps > 0 ? nsteps : 1), 0);
    int* hist_dev = nullptr;
    if (getenv("ASVD_DEBUG_HIST")) { ASVD_HIP_CHECK(hipMalloc(&hist_dev, 10 * sizeof(int))); }
    // 64x64 eigen-solves: the wave-local solver (evd_wave.hip) unless ASVD_EVDW=0 (the LDS solver of rounds 1-2, also used for the histogram)
    const bool evd_wave = !hist_dev && !(getenv("ASVD_EVDW") && atoi(getenv("ASVD_EVDW")) == 0);
    for (; sweep < max_sweeps; ++sweep) {
        const auto sweep_t0 = std::chrono::steady_clock::now();
        if (hist_dev) ASVD_HIP_CHECK(hipMemsetAsync(hist_dev, 0, 10 * sizeof(int), st));
        for (int g = 0; g < ngroups; ++g) {  // maxoff, nrot of this group's problems
            ASVD_HIP_CHECK(hipMemsetAsync(maxoff + gb0[g], 0, (size_t)gnb[g] * sizeof(int), gst[g]));
            ASVD_HIP_CHECK(hipMemsetAsync(nrot + gb0[g], 0, (size_t)gnb[g] * sizeof(int), gst[g]));
            ASVD_HIP_CHECK(hipMemsetAsync(nupd + gb0[g], 0, (size_t)gnb[g] * sizeof(int), gst[g]));
        }
        long long marked_total = 0;
        if (sparse) {
            // 1. snapshot of all couplings, per stream group
            for (int g = 0; g < ngroups; ++g) {
                ProfScope ps(5, gst[g]);
                const int b0 = gb0[g], nbg = gnb[g];
                const float* Xg = X + (int64_t)b0 * p.batch_stride;
                ASVD_HIP_CHECK(hipMemsetAsync(pflag + (size_t)b0 * p.nb * p.nb, 0, (size_t)nbg * p.nb * p.nb, gst[g]));
                panel_sumsq_kernel<<<dim3(p.nb, nbg), 256, 0, gst[g]>>>(sc, Xg, p.panel_stride, p.batch_stride, p.m_pad, p.n_pad,
                                                                        dnorm + (size_t)b0 * p.n_pad, done + b0);
                const unsigned nt = (unsigned)ceil_div64(p.nb, 4);
                static const int fc_tile = getenv("ASVD_SNAPSHOT_TILE") ? atoi(getenv("ASVD_SNAPSHOT_TILE")) : 128;  // 256 (fullcheck8_kernel) measured slower: 58.2 vs 56.1 ms for the three snapshots
                if (split_check && fc_tile == 256 && p.nb >= 16) {
                    static bool fc8_attr = false;
                    if (!fc8_attr) {
                        ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)fullcheck8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FC8_LDS_BYTES));
                        fc8_attr = true;
                    }
                    const unsigned nt8 = (unsigned)ceil_div64(p.nb, 8);
                    fullcheck8_kernel<<<dim3(nt8, nt8, nbg), 512, FC8_LDS_BYTES, gst[g]>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, p.m_pad, p.n_pad,
                                                                                         dnorm + (size_t)b0 * p.n_pad, tol, kb,
                                                                                         pflag + (size_t)b0 * p.nb * p.nb, maxoff + b0, done + b0);
                } else if (split_check)
                    fullcheck_kernel<1><<<dim3(nt, nt, nbg), 256, 0, gst[g]>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, p.m_pad, p.n_pad,
                                                                               dnorm + (size_t)b0 * p.n_pad, tol, kb,
                                                                               pflag + (size_t)b0 * p.nb * p.nb, maxoff + b0, done + b0);
                else
                    fullcheck_kernel<0><<<dim3(nt, nt, nbg), 256, 0, gst[g]>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, p.m_pad, p.n_pad,
                                                                               dnorm + (size_t)b0 * p.n_pad, tol, kb,
                                                                               pflag + (size_t)b0 * p.nb * p.nb, maxoff + b0, done + b0);
            }
            if (ngroups >= 2)
                for (int g = 0; g < ngroups; ++g) {
                    ASVD_HIP_CHECK(hipEventRecord(ev_join[g], gst[g]));
                    ASVD_HIP_CHECK(hipStreamWaitEvent(st, ev_join[g], 0));
                }
            hflag.resize((size_t)batch * p.nb * p.nb);
            ASVD_HIP_CHECK(hipMemcpyAsync(hflag.data(), pflag, hflag.size(), hipMemcpyDeviceToHost, st));
            ASVD_HIP_CHECK(hipStreamSynchronize(st));
            if (debug) fprintf(stderr, "[asvd_svd]   snapshot + readback %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sweep_t0).count());
            // 2. marks -> lists of disjoint pairs.  The pairs of one XOR step (I ^ J == d) are disjoint by construction, but late in the
            // iteration a step holds a handful of pairs and a sweep of nb-1 three-launch steps is all launch latency.  The marked
            // pairs are therefore packed first-fit, in schedule order (step, then I), into ROUNDS whose pairs share no panel (per
            // problem: a bitmask of used panels per round); a round is launched like a step.  Any order of disjoint rotations is a
            // valid Jacobi sweep; first-fit keeps the nearest-neighbour-first order of the XOR schedule among conflicting pairs.
            // ASVD_SPARSE_ROUNDS=0 keeps one list per XOR step.  (The list slots are indexed by `step` below in both cases.)
            hlist.clear();
            std::vector<std::vector<int>> tmp;  // [b_local * nsteps + round]
            const bool pack_rounds = !(getenv("ASVD_SPARSE_ROUNDS") && atoi(getenv("ASVD_SPARSE_ROUNDS")) == 0);
            const int words = (p.nb + 63) / 64;
            std::vector<uint64_t> used;  // [round][words] of the problem being packed
            for (int g = 0; g < ngroups; ++g) {
                const int b0 = gb0[g], nbg = gnb[g];
                tmp.assign((size_t)nbg * nsteps, std::vector<int>());
                for (int bl = 0; bl < nbg; ++bl) {
                    if (host_done[b0 + bl]) continue;
                    const unsigned char* f = hflag.data() + (size_t)(b0 + bl) * p.nb * p.nb;
                    bool packed = pack_rounds;
                    if (pack_rounds) {
                        used.assign((size_t)nsteps * words, 0);
                        long long cnt = 0;
                        for (int d = 1; d <= nsteps && packed; ++d)
                            for (int I = 0; I < p.nb && packed; ++I) {
                                const int J = I ^ d;
                                if (J <= I || J >= p.nb || !f[(size_t)I * p.nb + J]) continue;
                                int r = 0;
                                for (; r < nsteps; ++r) {
                                    uint64_t* u = &used[(size_t)r * words];
                                    if (!((u[I >> 6] >> (I & 63)) & 1) && !((u[J >> 6] >> (J & 63)) & 1)) {
                                        u[I >> 6] |= 1ull << (I & 63);
                                        u[J >> 6] |= 1ull << (J & 63);
                                        break;
                                    }
                                }
                                if (r == nsteps) { packed = false; break; }  // cannot happen below ~nsteps/2 marks per panel; fall back
                                tmp[(size_t)bl * nsteps + r].push_back((I << 16) | J);
                                ++cnt;
                            }
                        if (packed) marked_total += cnt;
                        else
                            for (int r = 0; r < nsteps; ++r) tmp[(size_t)bl * nsteps + r].clear();
                    }
                    if (!packed)
                        for (int I = 0; I < p.nb; ++I)
                            for (int J = I + 1; J < p.nb; ++J)
                                if (f[(size_t)I * p.nb + J]) { tmp[(size_t)bl * nsteps + ((I ^ J) - 1)].push_back((I << 16) | J); ++marked_total; }
                }
                for (int step = 0; step < nsteps; ++step) {
                    size_t mx = 0;
                    for (int bl = 0; bl < nbg; ++bl) mx = std::max(mx, tmp[(size_t)bl * nsteps + step].size());
                    sl_cnt[(size_t)g * nsteps + step] = (int)mx;
                    sl_off[(size_t)g * nsteps + step] = (int)hlist.size();
                    for (int bl = 0; bl < nbg; ++bl) {
                        const auto& v = tmp[(size_t)bl * nsteps + step];
                        for (size_t i = 0; i < mx; ++i) hlist.push_back(i < v.size() ? v[i] : -1);
                    }
                }
            }
            if (debug) fprintf(stderr, "[asvd_svd]   lists built at %.2f ms (%zu slots)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sweep_t0).count(), hlist.size());
            if (!hlist.empty()) ASVD_HIP_CHECK(hipMemcpyAsync(plist_dev, hlist.data(), hlist.size() * sizeof(int), hipMemcpyHostToDevice, st));
            if (ngroups >= 2) {
                ASVD_HIP_CHECK(hipEventRecord(ev_fork, st));
                for (int g = 0; g < ngroups; ++g) ASVD_HIP_CHECK(hipStreamWaitEvent(gst[g], ev_fork, 0));
            }
        }
        const int gstride = std::max(p.nsplit, p.fused ? p.nchunks_f : 0);  // partial-Gram slots per pair in the buffer
        // two-level dense sweep: only the internal step d = 1 runs through the single-level kernels (it also refreshes the carried
        // diagonal blocks); the super-steps follow below
        const bool two_now = p.two && !sparse;
        const int nsched = sparse ? nsteps : (two_now ? 1 : (int)sched.size());  // two-level: only the step d = 1 inside the super-panels
        for (int si = 0; si < nsched; ++si) {
            const int step = sparse ? si : (two_now ? si : sched[si]);
            for (int g = 0; g < ngroups; ++g) {
                const int b0 = gb0[g], nbg = gnb[g];
                hipStream_t s2 = gst[g];
                float* Xg = X + (int64_t)b0 * p.batch_stride;
                float* Gd32g = (float*)(wb + p.off_gd32) + (int64_t)b0 * p.nb * 1024;
                float* Gg = Gpart + (int64_t)b0 * p.npairs * gstride * 3072;
                float* Qg = Qbuf + (int64_t)b0 * p.npairs * PW * PW;
                int* ag = active + (int64_t)b0 * p.npairs;
                // fused path: the Gram blocks of this step were produced by the previous step's upgram launch (or by the last one
                // of the previous sweep); only the very first step of the call needs the stand-alone gram kernel
                if (sparse) {
                    const int slots = sl_cnt[(size_t)g * nsteps + step];
                    if (slots == 0) continue;  // no marked pair of this group meets in this step
                    const int* pl = plist_dev + sl_off[(size_t)g * nsteps + step];
                    // few pairs per launch: split the rows further to fill the CUs, within the partial-Gram capacity of a problem
                    const int64_t chunks = p.m_pad / 32, cap = (int64_t)p.npairs * p.nsplit / slots;
                    int64_t ns = std::max<int64_t>(p.nsplit, ceil_div64(768, (int64_t)slots * nbg));
                    ns = std::min<int64_t>(std::min<int64_t>(ns, cap), std::min<int64_t>(chunks, 64));
                    if (ns < 1) ns = 1;
                    const int rps = (int)(ceil_div64(chunks, ns) * 32);
                    const int nsp = (int)ceil_div64(p.m_pad, rps);
                    const int64_t iters = ceil_div64(p.R_upd, 128);
                    int64_t nc = std::min<int64_t>(iters, std::max<int64_t>(1, ceil_div64(1024, (int64_t)slots * nbg)));
                    const int rpw = (int)(ceil_div64(iters, nc) * 128);
                    const int nch = (int)ceil_div64(p.R_upd, rpw);
                    {
                        ProfScope ps(6, s2);
                        gram_kernel<<<dim3(nsp, slots, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, step, p.m_pad, rps, Gg,
                                                                           done + b0, pl, slots);
                    }
                    {
                        ProfScope ps(2, s2);
                        if (evd_wave)
                            launch_evdw0(false, slots, nbg, s2, sc, Gg, nsp, Qg, ag, maxoff + b0, nrot + b0, done + b0, tol, inner_sweeps, p.nb, step, kb,
                                         pl, slots, EvdV3{});
                        else
                            evd_kernel<0, 0><<<dim3(slots, nbg), 256, 0, s2>>>(sc, Gg, nsp, Qg, ag, maxoff + b0, nrot + b0, done + b0, tol, inner_sweeps,
                                                                               p.nb, step, kb, hist_dev, pl, slots, EvdV3{});
                    }
                    {
                        ProfScope ps(7, s2);
                        update_kernel<<<dim3(nch, slots, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, step, p.R_upd, rpw, Qg,
                                                                             ag, done + b0, pl, slots);
                    }
                    continue;
                }
                const bool gram_here = !p.fused || (sweep == 0 && step == 0);
                const int ns_here = gram_here ? p.nsplit : p.nchunks_f;
                if (gram_here) {
                    ProfScope ps(6, s2);
                    gram_kernel<<<dim3(p.nsplit, p.npairs, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, step, p.m_pad,
                                                                               p.rows_per_split, Gg, done + b0, nullptr, 0);
                }
                {
                    ProfScope ps(2, s2);
                    if (two_now) {  // internal step: also emits the fresh carried diagonal blocks of both panels of every pair
                        EvdV3 v3{};
                        v3.ns = p.ns;
                        v3.nbpan = p.nb;
                        v3.Gd32 = Gd32g;
                        if (evd_wave)
                            launch_evdw0(true, p.npairs, nbg, s2, sc, Gg, ns_here, Qg, ag, maxoff + b0, nrot + b0, done + b0, tol, inner_sweeps, p.nb, step,
                                         kb, nullptr, 0, v3);
                        else
                            evd_kernel<0, 1><<<dim3(p.npairs, nbg), 256, 0, s2>>>(sc, Gg, ns_here, Qg, ag, maxoff + b0, nrot + b0, done + b0, tol,
                                                                                   inner_sweeps, p.nb, step, kb, hist_dev, nullptr, 0, v3);
                    } else if (evd_wave) {
                        launch_evdw0(false, p.npairs, nbg, s2, sc, Gg, ns_here, Qg, ag, maxoff + b0, nrot + b0, done + b0, tol, inner_sweeps, p.nb, step, kb,
                                     nullptr, 0, EvdV3{});
                    } else {
                        evd_kernel<0, 0><<<dim3(p.npairs, nbg), 256, 0, s2>>>(sc, Gg, ns_here, Qg, ag, maxoff + b0, nrot + b0, done + b0, tol,
                                                                               inner_sweeps, p.nb, step, kb, hist_dev, nullptr, 0, EvdV3{});
                    }
                }
                {
                    ProfScope ps(7, s2);
                    if (p.fused) {
                        const int d = step + 1, e = (step + 1 < nsteps) ? step + 2 : 1;
                        upgram_kernel<<<dim3(p.nchunks_f, p.nb / 4, nbg), 256, 0, s2>>>(Xg, p.panel_stride, p.batch_stride, p.nb, d, e,
                                                                                        p.R_upd, p.m_pad, p.rows_per_wg_f, Qg, ag, Gg,
                                                                                        done + b0);
                    } else {
                        update_kernel<<<dim3(p.nchunks, p.npairs, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.nb, step,
                                                                                      p.R_upd, p.rows_per_wg, Qg, ag, done + b0, nullptr, 0);
                    }
                }
            }
        }
        // pipelined two-level sweep (twolevel.h "dual launches"): two halves of the batch, two phases apart, on ONE stream
        const bool piped = two_now && ngroups == 1 && batch >= 2 && split_piped && !super_rr_for(p) && !super_grouped_for(p);
        if (piped) {
            const int nsuper = 2 * p.npairs_s - 1;
            const int dup2 = std::min(nsuper, getenv("ASVD_DUP2") ? atoi(getenv("ASVD_DUP2")) : 0);
            std::vector<int> seq;
            for (int di = 0; di < nsuper + dup2; ++di) seq.push_back(di < dup2 ? di + 1 : di - dup2 + 1);
            const int L = (int)seq.size();
            const int hb0[2] = {0, (batch + 1) / 2}, hnb[2] = {(batch + 1) / 2, batch / 2};
            auto solve_args = [&](int h, int D) {
                SolveArgs a{};
                const int b0 = hb0[h];
                a.maxoff = maxoff + b0; a.nrot = nrot + b0; a.done = done + b0; a.tol = tol; a.inner_sweeps = inner_sweeps; a.nb = p.nb;
                a.step = D - 1; a.kb = kb; a.hist = hist_dev;
                a.v3.ns = p.ns; a.v3.nbpan = p.nb; a.v3.nsplit6 = p.nsplit_s;
                a.v3.Gx6 = (float*)(wb + p.off_gx6) + (int64_t)b0 * p.npairs_s * p.nsplit_s * 6 * 1024;
                a.v3.Gd32 = (float*)(wb + p.off_gd32) + (int64_t)b0 * p.nb * 1024;
                a.v3.Q0 = (float*)(wb + p.off_q0) + (int64_t)b0 * p.npairs_s * 2 * PW * PW;
                a.v3.D0 = (float*)(wb + p.off_d0) + (int64_t)b0 * p.npairs_s * 4 * 1024;
                a.v3.Qfin = (float*)(wb + p.off_qfin) + (int64_t)b0 * p.npairs_s * SP * SP;
                a.v3.subact = (int*)(wb + p.off_subact) + (int64_t)b0 * p.npairs_s * 4;
                a.gx = 2 * p.npairs_s; a.gy = hnb[h];
                return a;
            };
            auto gram_args = [&](int h, int D, int part) {  // part 0 / 1: lower / upper half of the pair slots; 2: all
                GramArgs ga{};
                const int b0 = hb0[h];
                ga.X = X + (int64_t)b0 * p.batch_stride; ga.panel_stride = p.panel_stride; ga.batch_stride = p.batch_stride; ga.ns = p.ns;
                ga.D = D; ga.m_pad = p.m_pad; ga.rows_per_split = p.rows_per_split_s;
                ga.Gx = (float*)(wb + p.off_gx6) + (int64_t)b0 * p.npairs_s * p.nsplit_s * 6 * 1024;
                ga.done = done + b0; ga.gx = p.nsplit_s; ga.gz = hnb[h]; ga.npairs = p.npairs_s;
                const int lo = p.npairs_s / 2;
                ga.pair0 = (part == 1) ? lo : 0;
                ga.gy = (part == 2) ? p.npairs_s : (part == 0 ? lo : p.npairs_s - lo);
                return ga;
            };
            auto launch_gs = [&](const SolveArgs& sa, const GramArgs& ga, int emode) {
                const int nblk = sa.gx * sa.gy + ga.gx * ga.gy * ga.gz;
                if (nblk == 0) return;
                ProfScope ps(emode ? 2 : 1, st);
                if (emode == 1) dual_gram_kernel<1><<<nblk, 256, 0, st>>>(sc, sa, ga);
                else dual_gram_kernel<2><<<nblk, 256, 0, st>>>(sc, sa, ga);
            };
            auto launch_u = [&](int h, int D) {
                const int b0 = hb0[h];
                ProfScope ps(3, st);
                supdate_split_kernel<<<dim3(p.nchunks_s, p.npairs_s, hnb[h]), 256, 0, st>>>(sc, 
                    X + (int64_t)b0 * p.batch_stride, p.panel_stride, p.batch_stride, p.ns, D, p.R_upd, p.rows_per_wg_s,
                    (float*)(wb + p.off_qfin) + (int64_t)b0 * p.npairs_s * SP * SP, (int*)(wb + p.off_subact) + (int64_t)b0 * p.npairs_s * 4,
                    done + b0, nupd + b0);
            };
            const GramArgs nog{};
            const SolveArgs nos{};
            launch_gs(nos, gram_args(1, seq[0], 2), 0);  // prologue: Gram tiles of the first step for half 1
            for (int i = 0; i < L; ++i) {
                const int D = seq[i];
                launch_gs(solve_args(1, D), gram_args(0, D, 0), 1);   // [Ga(h0) | E1(h1)]
                launch_gs(solve_args(1, D), gram_args(0, D, 1), 2);   // [Gb(h0) | E2(h1)]
                launch_u(1, D);
                if (i + 1 < L) {
                    launch_gs(solve_args(0, D), gram_args(1, seq[i + 1], 0), 1);   // [Ga'(h1) | E1(h0)]
                    launch_gs(solve_args(0, D), gram_args(1, seq[i + 1], 1), 2);   // [Gb'(h1) | E2(h0)]
                } else {
                    launch_gs(solve_args(0, D), nog, 1);
                    launch_gs(solve_args(0, D), nog, 2);
                }
                launch_u(0, D);
            }
        } else if (two_now) {
            const bool super_grp = super_grouped_for(p);
            const bool super_rr = super_rr_for(p) || super_grp;  // either way: no XOR structure at the super level (no fused update + Gram)
            const int nsuper = super_grp ? 15 + 16 * super_grouped_rounds(p) : (super_rr ? (p.ns + (p.ns & 1)) - 1 : 2 * p.npairs_s - 1);
            const bool split_bf16 = split_on;
            const bool gram_split = split_on && getenv("ASVD_GRAM_SPLIT") && atoi(getenv("ASVD_GRAM_SPLIT")) == 1;
            // local super-levels D = 1..L run twice at the start of the sweep (the two-level form of ASVD_DUP; ASVD_DUP2=L)
            const int dup2 = std::min(nsuper, getenv("ASVD_DUP2") ? atoi(getenv("ASVD_DUP2")) : 0);
            // supgram: the update of step D also leaves the Gram tiles of the step that follows (one pass instead of two); the
            // stand-alone Gram pass then runs only in front of the first super-step.  Needs split-bf16; ASVD_SUPGRAM=0 turns it off.
            // (round 3: also on the GROUPED schedule of counts that are a multiple of 16 — inside a group the steps are XOR steps, and the
            // 16 offsets of a group pair close quads {A_i, B_(i^s), B_(i^s'), A_(i^s^s')} as well; ASVD_SUPGRAM_GROUPED=0 keeps the separate passes)
            const bool fuse_grp = super_grp && !(getenv("ASVD_SUPGRAM_GROUPED") && atoi(getenv("ASVD_SUPGRAM_GROUPED")) == 0);
            const bool fuse_ug = split_bf16 && (!super_rr || fuse_grp) && p.npairs_s >= 2 && !(getenv("ASVD_SUPGRAM") && atoi(getenv("ASVD_SUPGRAM")) == 0);
            if (fuse_ug)
                ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)supgram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(SUPGRAM_SMEM_FLOATS * sizeof(float))));
            const int gx_slots = std::max(p.nsplit_s, p.nchunks_q);  // partial-tile slots per super-pair in the buffer
            // default: only on full schedules (ns a power of two).  Measured on the 13B shapes (ns = 80 in a 128-wide schedule):
            // always fused 21.4 s, fused where >= 75 % full 18.1 s, never fused 16.9 s for the model.
            const bool ns_pow2 = (p.ns & (p.ns - 1)) == 0;
            const double fill_min = getenv("ASVD_SUPGRAM_FILL") ? atof(getenv("ASVD_SUPGRAM_FILL")) : (ns_pow2 ? 1.0 : 2.0);
            auto level_of = [&](int di) { return di < dup2 ? di + 1 : di - dup2 + 1; };
            for (int di = 0; di < nsuper + dup2; ++di) {
                const int D = level_of(di);
                const int E = (di + 1 < nsuper + dup2) ? level_of(di + 1) : 0;  // 0: last super-step of the sweep
                // supgram reads EVERY panel of a quad with a present member; on a padded schedule (ns not a power of two) many super-steps
                // hold few real pairs, and there the separate passes, which only touch the pairs that exist, move fewer bytes:
                // fuse when the two steps together hold >= fill_min of the pair slots a full schedule would
                auto real_pairs = [&](int d) { int n = 0; for (int S = 0; S < p.ns; ++S) n += ((S ^ d) > S && (S ^ d) < p.ns) ? 1 : 0; return n; };
                auto fused_after = [&](int dj) {  // does the launch of super-step index dj also leave the tiles of index dj + 1 ?
                    if (!fuse_ug || dj < 0 || dj + 1 >= nsuper + dup2) return false;
                    if (super_grp) {  // consecutive steps inside the groups (XOR distances 1..15), or consecutive offsets of the same round of group pairs
                        const int s0 = level_of(dj) - 1, s1 = level_of(dj + 1) - 1;   // 0-based super-steps
                        if (s1 != s0 + 1) return false;
                        return s1 < 15 || (s0 >= 15 && ((s0 - 15) >> 4) == ((s1 - 15) >> 4));
                    }
                    const int d0 = level_of(dj), d1 = level_of(dj + 1);
                    return d0 != d1 && (double)(real_pairs(d0) + real_pairs(d1)) >= fill_min * 2.0 * (p.ns / 2);
                };
                const bool gram_in = !fused_after(di - 1);   // tiles of this step not left by the previous launch
                const bool gram_out = fused_after(di);
                for (int g = 0; g < ngroups; ++g) {
                    const int b0 = gb0[g], nbg = gnb[g];
                    hipStream_t s2 = gst[g];
                    float* Xg = X + (int64_t)b0 * p.batch_stride;
                    EvdV3 v3{};
                    v3.ns = p.ns;
                    v3.nbpan = p.nb;
                    v3.nsplit6 = gram_in ? p.nsplit_s : p.nchunks_q;
                    float* Gx6g = (float*)(wb + p.off_gx6) + (int64_t)b0 * p.npairs_s * gx_slots * 6 * 1024;
                    v3.Gx6 = Gx6g;
                    v3.Gd32 = (float*)(wb + p.off_gd32) + (int64_t)b0 * p.nb * 1024;
                    v3.Q0 = (float*)(wb + p.off_q0) + (int64_t)b0 * p.npairs_s * 2 * PW * PW;
                    v3.D0 = (float*)(wb + p.off_d0) + (int64_t)b0 * p.npairs_s * 4 * 1024;
                    v3.Qfin = (float*)(wb + p.off_qfin) + (int64_t)b0 * p.npairs_s * SP * SP;
                    v3.subact = (int*)(wb + p.off_subact) + (int64_t)b0 * p.npairs_s * 4;
                    if (gram_in) {
                        ProfScope ps(1, s2);
                        if (gram_split)
                            sgram6_kernel<1><<<dim3(p.nsplit_s, p.npairs_s, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.ns, D, p.m_pad,
                                                                                               p.rows_per_split_s, Gx6g, done + b0);
                        else
                            sgram6_kernel<0><<<dim3(p.nsplit_s, p.npairs_s, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.ns, D, p.m_pad,
                                                                                               p.rows_per_split_s, Gx6g, done + b0);
                    }
                    {
                        ProfScope ps(2, s2);
                        if (evd_wave) {  // both inner steps of every super-pair in one launch, one wave per 64x64 solve (evd_wave.hip)
                            launch_evdw12(p.npairs_s, nbg, s2, sc, maxoff + b0, nrot + b0, done + b0, tol, inner_sweeps, p.nb, D - 1, kb, v3);
                        } else {
                            evd_kernel<1, 1><<<dim3(2 * p.npairs_s, nbg), 256, 0, s2>>>(sc, nullptr, 0, nullptr, nullptr, maxoff + b0, nrot + b0, done + b0,
                                                                                       tol, inner_sweeps, p.nb, D - 1, kb, hist_dev, nullptr, 0, v3);
                            evd_kernel<2, 1><<<dim3(2 * p.npairs_s, nbg), 256, 0, s2>>>(sc, nullptr, 0, nullptr, nullptr, maxoff + b0, nrot + b0, done + b0,
                                                                                       tol, inner_sweeps, p.nb, D - 1, kb, hist_dev, nullptr, 0, v3);
                        }
                    }
                    {
                        ProfScope ps(gram_out ? 8 : 3, s2);
                        if (gram_out)
                            supgram_kernel<<<dim3(p.nchunks_q, (2 * p.npairs_s) / 4, nbg), 512, SUPGRAM_SMEM_FLOATS * sizeof(float), s2>>>(sc, 
                                Xg, p.panel_stride, p.batch_stride, p.ns, D, E, p.R_upd, p.m_pad, p.rows_per_wg_q, v3.Qfin, v3.subact, Gx6g, done + b0,
                                nupd + b0, p.npairs_s);
                        else if (split_bf16)
                            supdate_split_kernel<<<dim3(p.nchunks_s, p.npairs_s, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.ns, D,
                                                                                                    p.R_upd, p.rows_per_wg_s, v3.Qfin, v3.subact, done + b0, nupd + b0);
                        else
                            supdate_kernel<<<dim3(p.nchunks_s, p.npairs_s, nbg), 256, 0, s2>>>(sc, Xg, p.panel_stride, p.batch_stride, p.ns, D, p.R_upd,
                                                                                              p.rows_per_wg_s, v3.Qfin, v3.subact, done + b0, nupd + b0);
                    }
                }
            }
        }
        if (ngroups >= 2) {
            for (int g = 0; g < ngroups; ++g) {
                ASVD_HIP_CHECK(hipEventRecord(ev_join[g], gst[g]));
                ASVD_HIP_CHECK(hipStreamWaitEvent(st, ev_join[g], 0));
            }
        }
        ASVD_HIP_CHECK(hipMemcpyAsync(flags.data(), wb + p.off_flags, (size_t)batch * 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        ASVD_HIP_CHECK(hipStreamSynchronize(st));
        if (hist_dev) {
            int hh[10];
            ASVD_HIP_CHECK(hipMemcpy(hh, hist_dev, sizeof(hh), hipMemcpyDeviceToHost));
            fprintf(stderr, "[asvd_svd] sweep %d pair-measure histogram by decade 1e0..1e-9:", sweep + 1);
            for (int i = 0; i < 10; ++i) fprintf(stderr, " %d", hh[i]);
            fprintf(stderr, "\n");
        }
        if (g_prof_enabled) {
            long long rot = 0;
            for (int b = 0; b < batch; ++b) rot += host_done[b] ? 0 : flags[batch + b];
            g_prof_sweep_ms.push_back((float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sweep_t0).count());
            g_prof_sweep_rot.push_back(rot);
        }
        if (debug) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sweep_t0).count();
            long rot = 0;
            for (int b = 0; b < batch; ++b) rot += host_done[b] ? 0 : flags[batch + b];
            fprintf(stderr, "[asvd_svd] sweep %d%s wall %.2f ms, rotated pairs (all problems) %ld, marked %lld\n", sweep + 1, sparse ? " (sparse)" : "", ms, rot, marked_total);
        }
        bool all_done = true;
        bool changed = false;
        for (int b = 0; b < batch; ++b) {
            if (host_done[b]) continue;
            float mo;
            unsigned bits = (unsigned)flags[b];
            std::memcpy(&mo, &bits, sizeof(float));
            sweeps_done[b] = sweep + 1;
            last_rot[b] = flags[batch + b];
            if (g_prof_enabled) { g_prof_pairs[0] += sparse ? 0 : (long long)p.nb * (p.nb - 1) / 2; g_prof_pairs[1] += last_rot[b]; g_prof_pairs[2] += flags[3 * batch + b]; }
            last_off[b] = mo;
            if (debug) fprintf(stderr, "[asvd_svd] b=%d sweep=%d maxoff=%.3e rotated_pairs=%d\n", b, sweep + 1, mo, last_rot[b]);
            if (mo != mo) { status[b] = ASVD_N_NAN; host_done[b] = 1; changed = true; }
            // Jacobi converges quadratically: a sweep that STARTED with max |cos| = mo leaves ~mo^2 behind, so a sweep with
            // mo < 0.3 sqrt(tol) has already produced orthogonality below tol (pairs above tol were all rotated in it).
            else if (mo < tol || mo < 0.3f * sqrtf(tol)) { status[b] = ASVD_OK; host_done[b] = 1; changed = true; }
            // stagnation at the fp32 noise floor: the scaled off-diagonal stopped contracting well below the level that
            // matters for the 1e-4 sigma / 1e-3 reconstruction contract (second-order in maxoff) -> converged
            else if (mo < 100.0f * tol && mo > 0.25f * prev_off[b]) { status[b] = ASVD_OK; host_done[b] = 1; changed = true; }
            else all_done = false;
            prev_off[b] = mo;
        }
        if (g_prof_enabled && sparse) g_prof_pairs[0] += marked_total;
        if (all_done) { ++sweep; break; }
        {
            // the next sweep is sparse once fewer than half of the pairs of the still-running problems rotated in this one
            long long rot = 0, tot = 0;
            for (int b = 0; b < batch; ++b)
                if (!host_done[b]) { rot += last_rot[b]; tot += (long long)p.nb * (p.nb - 1) / 2; }
            sparse = sparse_allowed && tot > 0 && (double)rot < sparse_frac * (double)tot;
        }
        if (changed) {
            ASVD_HIP_CHECK(hipMemcpyAsync(done, host_done.data(), (size_t)batch * sizeof(int), hipMemcpyHostToDevice, st));
            ASVD_HIP_CHECK(hipStreamSynchronize(st));  // host_done may be modified next sweep
        }
    }

    if (hist_dev) (void)hipFree(hist_dev);
    if (ngroups >= 2) {
        (void)hipEventDestroy(ev_fork);
        for (int g = 0; g < ngroups; ++g) (void)hipEventDestroy(ev_join[g]);
    }

    // ---- finalize ----
    {
        ProfScope ps(4, st);
        if (p.vmode == 2) {
            const int64_t ops = (int64_t)p.m_pad * PB;
            backsolve_kernel<<<dim3((unsigned)ceil_div64(p.nb, 4), (unsigned)ceil_div64(p.nb, 4), batch), 256, 0, st>>>(
                (const float*)(wb + p.off_xorig), ops, ops * p.nb, X, p.panel_stride, p.batch_stride, p.nb, p.m_pad);
        }
        colnorm_kernel<<<dim3(p.nb, batch), 256, 0, st>>>(X, p.panel_stride, p.batch_stride, p.m_pad, p.R, p.n_pad,
                                                          p.vmode == 1 ? 1 : 0, sig, ina, inv);
        rank_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(sig, p.n_pad, perm);
        for (int b = 0; b < batch; ++b) {
            float* Uo = U_host ? U_host[b] : nullptr;
            float* Vo = V_host ? V_host[b] : nullptr;
            // oriented left vectors (A part) are U of A when not transposed, V of A when transposed
            float* outA = p.transposed ? Vo : Uo;
            float* outV = p.transposed ? Uo : Vo;
            const int rowsA = outA ? p.rows : 0;
            const int rowsV = (p.vmode != 0 && outV) ? p.cols : 0;   // V rows exist only when the right vectors were requested
            dim3 grid((unsigned)ceil_div64(k, 64), (unsigned)ceil_div64(rowsA + rowsV > 0 ? rowsA + rowsV : 1, 64));
            gather_kernel<<<grid, 256, 0, st>>>(X + (int64_t)b * p.batch_stride, p.panel_stride, p.m_pad, p.R,
                                                sig + (int64_t)b * p.n_pad, ina + (int64_t)b * p.n_pad,
                                                inv + (int64_t)b * p.n_pad, perm + (int64_t)b * p.n_pad, rowsA, rowsV,
                                                (int)k, outA, outV, S_host[b]);
        }
    }
    ASVD_HIP_CHECK(hipStreamSynchronize(st));
    ASVD_HIP_CHECK(hipGetLastError());
    if (g_prof_enabled && manage_profile) prof_end();

    int worst = ASVD_OK;
    for (int b = 0; b < batch; ++b) {
        if (info_host) {
            info_host[4 * b + 0] = status[b];
            info_host[4 * b + 1] = sweeps_done[b];
            info_host[4 * b + 2] = last_rot[b];
            std::memcpy(&info_host[4 * b + 3], &last_off[b], sizeof(float));  // last sweep's max scaled off-diagonal (float bits)
        }
        if (status[b] > worst) worst = status[b];
    }
    return worst;
}


// ---------------------------------------------------------------------------------------------------------------------
// tall path driver (see the kernel block "Tall problems" above)
static bool tall_wanted(const Plan& p) {
    // The reduction pays for every shape: for tall problems it shrinks each Jacobi step from rows x cols to cols x cols, and for all
    // of them the norm-sorted Cholesky-QR is a preconditioner (Jacobi on R^T: 14 -> 10 sweeps at 4096^2, better orthogonality).
    if (getenv("ASVD_NO_REDUCE")) return false;
    return p.cols >= 128;
}

struct TallLayout {
    size_t off_xp, off_g, off_gs, off_dp, off_df, off_perm, off_dg, off_d, off_fail, off_r, off_vr, off_part, off_inv, off_inner, inner_bytes, total;
    int n_pad64;
};

static int tall_layout(int batch, const Plan& p, int want_vectors, int64_t k, TallLayout& t) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    t.n_pad64 = p.n_pad;  // n_pad is a multiple of 64 = Cholesky block
    t.off_xp = take((size_t)p.m_pad * PB * p.nb * batch * sizeof(float));
    t.off_g = take((size_t)p.n_pad * p.n_pad * batch * sizeof(double));
    t.off_gs = take((size_t)p.n_pad * p.n_pad * batch * sizeof(double));
    t.off_dp = take((size_t)p.n_pad * batch * sizeof(double));
    t.off_df = take((size_t)p.n_pad * batch * sizeof(float));
    t.off_perm = take((size_t)p.n_pad * batch * sizeof(int));
    t.off_dg = take((size_t)(p.n_pad / CB) * CB * CB * batch * sizeof(double));
    t.off_d = take((size_t)p.n_pad * batch * sizeof(double));
    t.off_fail = take((size_t)batch * sizeof(int));
    t.off_r = take((size_t)p.n_pad * p.n_pad * batch * sizeof(float));
    t.off_vr = take(want_vectors ? (size_t)2 * p.cols * k * batch * sizeof(float) : 0);  // permuted + un-permuted right vectors
    t.off_part = take(want_vectors ? (size_t)64 * k * batch * sizeof(double) : 0);  // per problem: the epilogues run on side streams
    t.off_inv = take(want_vectors ? (size_t)k * batch * sizeof(float) : 0);
    Plan pi;
    int rc = make_plan(batch, p.cols, p.cols, want_vectors, want_vectors, pi);
    if (rc) return rc;
    t.inner_bytes = pi.total;
    t.off_inner = take(pi.total);
    t.total = off;
    return ASVD_OK;
}

static size_t tall_worksize(int batch, int64_t m, int64_t n, int want_vectors, int64_t k) {
    Plan p;
    if (make_plan(batch, m, n, 0, 0, p)) return 0;
    TallLayout t;
    if (tall_layout(batch, p, want_vectors, k, t)) return 0;
    return t.total;
}

// returns ASVD_OK.., or -100 when the reduction is not applicable (Cholesky breakdown): the caller then runs the direct path
static int svd_tall(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda, const void* const* cs_host,
                    int cs_dtype, float* const* U_host, float* const* S_host, float* const* V_host, int64_t k, int max_sweeps,
                    float tol, void* work, size_t work_bytes, int* info_host, void* stream) {
    Plan p;
    int rc = make_plan(batch, m, n, 0, 0, p);  // panels hold the A rows only
    if (rc) return rc;
    const int want_vectors = (U_host || V_host) ? 1 : 0;
    TallLayout t;
    rc = tall_layout(batch, p, want_vectors, k, t);
    if (rc) return rc;
    if (work_bytes < t.total) return -100;
    hipStream_t st = (hipStream_t)stream;
    char* wb = (char*)work;
    float* Xp = (float*)(wb + t.off_xp);
    double* G = (double*)(wb + t.off_g);
    double* d = (double*)(wb + t.off_d);
    double* Dg = (double*)(wb + t.off_dg);
    double* Gs = (double*)(wb + t.off_gs);
    double* dp = (double*)(wb + t.off_dp);
    float* dF = (float*)(wb + t.off_df);
    int* cperm = (int*)(wb + t.off_perm);
    const bool sort_cols = true;  // norm-sorted column order before the Cholesky factorisation (14 -> 10 sweeps; unsorted saves nothing)
    int* fail = (int*)(wb + t.off_fail);
    float* R = (float*)(wb + t.off_r);
    float* Vr = (float*)(wb + t.off_vr);
    const int64_t ldg = p.n_pad, gbs = (int64_t)p.n_pad * p.n_pad;
    const int nbk = p.n_pad / CB;
    // Jacobi on R^T (default): the leading right vectors of X are then the directly rotated columns (orthogonal to 1e-6) instead of
    // backsolved ones, which matters because the long-side vectors X v / sigma amplify the error of v by sigma_1 / sigma_j;
    // row-scaled problems (wide layers: the activation scales sit on the long side) also need 3 fewer sweeps.
    const bool use_rt = true;

    {
        ProfScope ps(0, st);
        ASVD_HIP_CHECK(hipMemsetAsync(Xp, 0, (size_t)p.batch_stride * batch * sizeof(float), st));
        ASVD_HIP_CHECK(hipMemsetAsync(fail, 0, (size_t)batch * sizeof(int), st));
        for (int b = 0; b < batch; ++b) {
            float* Xb = Xp + (int64_t)b * p.batch_stride;
            const void* sc = cs_host ? cs_host[b] : nullptr;
            int prc;
            switch (a_dtype) {
                case ASVD_F32: prc = launch_pack<ASVD_F32>(a_host[b], lda, sc, cs_dtype, p, Xb, st); break;
                case ASVD_F16: prc = launch_pack<ASVD_F16>(a_host[b], lda, sc, cs_dtype, p, Xb, st); break;
                default: prc = launch_pack<ASVD_BF16>(a_host[b], lda, sc, cs_dtype, p, Xb, st); break;
            }
            if (prc) return prc;
        }
        gram64_kernel<<<dim3(p.nb, (unsigned)ceil_div64(p.nb, 4), batch), 256, 0, st>>>(Xp, p.panel_stride, p.batch_stride, p.nb, p.m_pad, G, ldg, gbs);
        chol_diag_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(G, ldg, gbs, p.n_pad, d);
        // sort columns by decreasing norm (stable, padding last), permute + unit-scale the Gram matrix
        d_to_float_kernel<<<(unsigned)ceil_div64((int64_t)p.n_pad * batch, 256), 256, 0, st>>>(d, p.n_pad * batch, dF);
        if (sort_cols) rank_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(dF, p.n_pad, cperm);
        else iota_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(cperm, p.n_pad);
        g_permute_scale_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), p.n_pad, batch), 256, 0, st>>>(G, ldg, gbs, d, cperm, p.n_pad, Gs, dp);
        // block rows in groups of `cg`: inside a group each finished row updates the rest of the group's strip (K = 64), the matrix behind
        // the group is updated once per group (K = 64 cg)
        static const int cg = std::max(1, getenv("ASVD_CHOL_GROUP") ? atoi(getenv("ASVD_CHOL_GROUP")) : 4);
        for (int j0 = 0; j0 < nbk; j0 += cg) {
            const int j1 = std::min(nbk, j0 + cg);
            for (int jb = j0; jb < j1; ++jb) {
                static const bool panel_wave = !(getenv("ASVD_CHOL_PANEL_WAVE") && atoi(getenv("ASVD_CHOL_PANEL_WAVE")) == 0);
                if (panel_wave) {
                    chol_diag_wave_kernel<<<batch, 64, 0, st>>>(Gs, ldg, gbs, jb, fail, Dg, nbk);
                    if (jb + 1 < nbk) chol_trsm_kernel<<<dim3(nbk - jb - 1, batch), 256, 0, st>>>(Gs, ldg, gbs, jb);
                }
                else chol_panel_kernel<<<dim3(std::min(nbk - jb, 16), batch), 256, 0, st>>>(Gs, ldg, gbs, jb, fail, Dg, nbk);
                if (jb + 1 < j1) chol_syrk_kernel<<<dim3(j1 - jb - 1, nbk - jb - 1, batch), 256, 0, st>>>(Gs, ldg, gbs, jb, nbk);
            }
            if (j1 < nbk) chol_syrk_multi_kernel<<<dim3(nbk - j1, nbk - j1, batch), 256, 0, st>>>(Gs, ldg, gbs, j0, j1 - j0, nbk);
        }
        if (use_rt)  // n_pad is a multiple of 64
            r_to_f32_t_kernel<<<dim3((unsigned)(p.n_pad / 32), (unsigned)(p.n_pad / 32), batch), 256, 0, st>>>(Gs, ldg, gbs, Dg, dp, p.n_pad, R, gbs);
        else
            r_to_f32_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), p.n_pad, batch), 256, 0, st>>>(Gs, ldg, gbs, Dg, dp, p.n_pad, 0, R, gbs);
    }
    std::vector<int> hfail(batch, 0);
    ASVD_HIP_CHECK(hipMemcpyAsync(hfail.data(), fail, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, st));
    ASVD_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < batch; ++b)
        if (hfail[b]) {
            if (getenv("ASVD_DEBUG")) fprintf(stderr, "[asvd_svd] Cholesky-QR breakdown for problem %d (code %d): falling back to the direct path\n", b, hfail[b]);
            return -100;
        }

    // right vectors of X = right vectors of R  (left vectors of R^T when ASVD_RT): written straight to the caller's short-side
    // output when it exists, else to the temporary
    std::vector<const void*> rp(batch);
    std::vector<float*> vr(batch, nullptr), vperm(batch, nullptr);
    for (int b = 0; b < batch; ++b) {
        rp[b] = R + (int64_t)b * gbs;
        if (want_vectors) {
            float* short_out = p.transposed ? (U_host ? U_host[b] : nullptr) : (V_host ? V_host[b] : nullptr);
            vperm[b] = Vr + (int64_t)(2 * b) * p.cols * k;                       // right vectors in sorted-column order
            vr[b] = short_out ? short_out : Vr + (int64_t)(2 * b + 1) * p.cols * k;  // ... in the original order
        }
    }
    float* const* inner_U = nullptr;
    float* const* inner_V = nullptr;
    if (want_vectors) { if (use_rt) inner_U = vperm.data(); else inner_V = vperm.data(); }
    rc = svd_direct(batch, rp.data(), ASVD_F32, p.cols, p.cols, p.n_pad, nullptr, 0, inner_U, S_host, inner_V, k, max_sweeps, tol,
                    wb + t.off_inner, t.inner_bytes, info_host, stream, false);
    if (rc < 0) return rc;
    if (want_vectors) {
        ProfScope ps(4, st);
        // the per-problem epilogues (un-permute, long-side GEMM, sigma refinement) are independent: spread them over three side
        // streams so the 1024-workgroup GEMMs overlap each other's tails
        constexpr int NES = 3;
        constexpr int MAXDEV_E = 16;
        static hipStream_t s_epi_dev[MAXDEV_E][NES] = {};
        int devid_e = 0;
        ASVD_HIP_CHECK(hipGetDevice(&devid_e));
        if (devid_e < 0 || devid_e >= MAXDEV_E) return ASVD_E_BADARG;
        hipStream_t* s_epi = s_epi_dev[devid_e];
        hipEvent_t e_fork = nullptr, e_join[NES] = {nullptr, nullptr, nullptr};
        const int nes = (batch >= 2 && getenv("ASVD_EPI_STREAMS") && atoi(getenv("ASVD_EPI_STREAMS")) > 1) ? NES : 1;  // one stream by default: see common.h
        if (nes > 1) {
            ASVD_HIP_CHECK(hipEventCreateWithFlags(&e_fork, hipEventDisableTiming));
            ASVD_HIP_CHECK(hipEventRecord(e_fork, st));
            for (int i = 0; i < nes; ++i) {
                if (!s_epi[i]) ASVD_HIP_CHECK(hipStreamCreateWithFlags(&s_epi[i], hipStreamNonBlocking));
                ASVD_HIP_CHECK(hipStreamWaitEvent(s_epi[i], e_fork, 0));
                ASVD_HIP_CHECK(hipEventCreateWithFlags(&e_join[i], hipEventDisableTiming));
            }
        }
        const bool gemm_split = !(getenv("ASVD_SPLIT") && atoi(getenv("ASVD_SPLIT")) == 0);  // split-bf16 arithmetic, as in the sweeps
        for (int b = 0; b < batch; ++b) {
            hipStream_t se = nes > 1 ? s_epi[b % nes] : st;
            row_unpermute_kernel<<<dim3((unsigned)ceil_div64(k, 256), p.cols), 256, 0, se>>>(vperm[b], cperm + (int64_t)b * p.n_pad, p.cols, (int)k, vr[b]);
            float* long_out = p.transposed ? (V_host ? V_host[b] : nullptr) : (U_host ? U_host[b] : nullptr);
            if (!long_out) continue;
            if (gemm_split)
                nn_gemm_split_kernel<<<dim3((unsigned)ceil_div64(k, 128), (unsigned)ceil_div64(p.rows, 128)), 256, 0, se>>>(
                    Xp + (int64_t)b * p.batch_stride, p.panel_stride, p.nb, p.rows, p.cols, vr[b], k, nullptr, (int)k, long_out, k);
            else
                nn_gemm_kernel<<<dim3((unsigned)ceil_div64(k, 128), (unsigned)ceil_div64(p.rows, 128)), 256, 0, se>>>(
                    Xp + (int64_t)b * p.batch_stride, p.panel_stride, p.nb, p.rows, p.cols, vr[b], k, nullptr, (int)k, long_out, k);
            // sigma_j = |X v_j| and unit left vectors
            const int nsp = (int)std::min<int64_t>(64, ceil_div64(p.rows, 256));
            const int rps = (int)ceil_div64(p.rows, nsp);
            double* part = (double*)(wb + t.off_part) + (size_t)b * 64 * k;
            float* invs = (float*)(wb + t.off_inv) + (size_t)b * k;
            colsumsq_kernel<<<dim3((unsigned)ceil_div64(k, 64), nsp), 256, 0, se>>>(long_out, k, p.rows, (int)k, rps, part);
            colfinish_kernel<<<1, 256, 0, se>>>(part, nsp, (int)k, S_host[b], invs);
            colscale_kernel<<<dim3((unsigned)ceil_div64(k, 256), (unsigned)ceil_div64(p.rows, 32)), 256, 0, se>>>(long_out, k, p.rows, (int)k, invs);
        }
        if (nes > 1) {
            for (int i = 0; i < nes; ++i) {
                ASVD_HIP_CHECK(hipEventRecord(e_join[i], s_epi[i]));
                ASVD_HIP_CHECK(hipStreamWaitEvent(st, e_join[i], 0));
            }
        }
        ASVD_HIP_CHECK(hipStreamSynchronize(st));
        ASVD_HIP_CHECK(hipGetLastError());
        if (nes > 1) {
            (void)hipEventDestroy(e_fork);
            for (int i = 0; i < nes; ++i) (void)hipEventDestroy(e_join[i]);
        }
    }
    return rc;
}

// Test hook (tests/test_gpu_twolevel.py): ONE launch of the two-level update kernel on caller-built panels, so that the kernel can
// be checked against a plain fp64 product.  X: [batch][nb][R][32] fp32 panels; Qfin: [batch][npairs][128*128]; subact: [batch][npairs][4];
// done, nupd: [batch] ints (all device pointers).  split != 0 selects the split-bf16 kernel.
int asvd_test_supdate(int split, float* X, int64_t panel_stride, int64_t batch_stride, int ns, int D, int R, int rows_per_wg, const float* Qfin,
                      const int* subact, const int* done, int* nupd, int nchunks, int npairs, int batch, void* stream) {
    if (!X || !Qfin || !subact || !done || !nupd || ns < 2 || D < 1 || R < 32 || (R % 32) || rows_per_wg < 32 || (rows_per_wg % 32)) return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const Sched sc = default_sched();
    if (split)
        supdate_split_kernel<<<dim3(nchunks, npairs, batch), 256, 0, st>>>(sc, X, panel_stride, batch_stride, ns, D, R, rows_per_wg, Qfin, subact, done, nupd);
    else
        supdate_kernel<<<dim3(nchunks, npairs, batch), 256, 0, st>>>(sc, X, panel_stride, batch_stride, ns, D, R, rows_per_wg, Qfin, subact, done, nupd);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

// Test hook: the super-panel pair schedule itself.  out[step * npairs + k] = (S << 16) | T of slot k of super-step `step`, or -1 for an empty
// slot, for the order the library would pick for `ns` super-panels (1 XOR, 2 grouped; see super_pair).  Returns the number of super-steps
// through *nsteps_out.  tests/test_gpu_twolevel.py checks that every pair appears exactly once and that the pairs of a step are disjoint.
}  // extern "C"
namespace {
__global__ void super_schedule_kernel(Sched sc, int ns, int nsteps, int npairs, int* __restrict__ out) {
    const int step = blockIdx.x, k = threadIdx.x;
    if (k >= npairs) return;
    int S, T;
    super_pair(sc, ns, step, k, S, T);
    out[step * npairs + k] = (S < ns && T < ns) ? ((S << 16) | T) : -1;
}
}  // namespace
extern "C" {
int asvd_test_super_schedule(int ns, int grouped, int* out_dev, int out_capacity, int* nsteps_out, int* npairs_out) {
    if (ns < 2 || ns > 1024 || !out_dev || !nsteps_out || !npairs_out) return ASVD_E_BADARG;
    int pw2 = 2;
    while (pw2 < ns) pw2 <<= 1;
    const int npairs = pw2 / 2;
    const bool grp = grouped && grouped_applies(ns);
    const int ng = ns / 16;
    const int nsteps = grp ? 15 + 16 * ((ng & 1) ? ng : ng - 1) : pw2 - 1;
    *nsteps_out = nsteps;
    *npairs_out = npairs;
    if ((int64_t)nsteps * npairs > out_capacity || npairs > 1024) return ASVD_E_WORKSPACE;
    Sched sc = default_sched();
    sc.super_order = grp ? 2 : 1;
    if (grp) set_group_table(sc, ns);
    super_schedule_kernel<<<nsteps, npairs>>>(sc, ns, nsteps, npairs, out_dev);
    ASVD_HIP_CHECK(hipGetLastError());
    ASVD_HIP_CHECK(hipDeviceSynchronize());
    return ASVD_OK;
}

// Test hook: ONE launch of the fused update + next-step Gram kernel (supgram_kernel) on caller-built panels.  Gx: [batch][npairs][nchunks][6][1024]
// partial tiles of super-step E, indexed by E's pair slots.  E != D, both in [1, 2 * npairs).
int asvd_test_supgram(float* X, int64_t panel_stride, int64_t batch_stride, int ns, int D, int E, int R, int m_pad, int rows_per_wg,
                      const float* Qfin, const int* subact, float* Gx, const int* done, int* nupd, int nchunks, int npairs, int batch, void* stream) {
    if (!X || !Qfin || !subact || !Gx || !done || !nupd || ns < 3 || npairs < 2 || D < 1 || E < 1 || D == E || D >= 2 * npairs || E >= 2 * npairs ||
        R < 32 || (R % 32) || (m_pad % 32) || rows_per_wg < 32 || (rows_per_wg % 32))
        return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Sched sc = default_sched();
    if (getenv("ASVD_SG_ABLATE")) sc.dbg_fill = atoi(getenv("ASVD_SG_ABLATE")) << 8;  // tools/bench_supgram.py
    ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)supgram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SUPGRAM_SMEM_FLOATS * sizeof(float))));
    supgram_kernel<<<dim3(nchunks, (2 * npairs) / 4, batch), 512, SUPGRAM_SMEM_FLOATS * sizeof(float), st>>>(sc, X, panel_stride, batch_stride, ns, D, E, R, m_pad,
                                                                                                           rows_per_wg, Qfin, subact, Gx, done, nupd, npairs);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

#ifdef ASVD_SG_TIMING
int asvd_test_sg_timing(unsigned long long* out_host) {  // [2][64][10] s_memtime stamps of the last supgram launch (timing builds only)
    ASVD_HIP_CHECK(hipDeviceSynchronize());
    ASVD_HIP_CHECK(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sg_ts), 2 * 64 * 10 * sizeof(unsigned long long)));
    return ASVD_OK;
}
#endif

// Test hook (tests/test_gpu_evd_wave.py): the wave-local 64x64 eigen-solver alone.  G: [batch][64][64] symmetric (device); outputs
// Q [batch][64][64] (unsorted, unscaled), diag / rnk / cs [batch][64], Gout [batch][64][64] (the image the sweeps leave), meas [batch][2].
int asvd_test_evd_wave(const float* G, int batch, int sweeps, float* Q, float* diag, int* rnk, float* cs, float* Gout, float* meas, void* stream) {
    if (!G || !Q || !diag || !rnk || !cs || !Gout || !meas || batch < 1 || sweeps < 0) return ASVD_E_BADARG;
    launch_evdw_test(batch, (hipStream_t)stream, G, sweeps, Q, diag, rnk, cs, Gout, meas);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

int asvd_svd_batched(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                     const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                     float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                     int* info_host, void* stream) {
    if (!a_host || !S_host || !work || !dtype_ok(a_dtype) || lda < n) return ASVD_E_BADARG;
    if (cs_host && !dtype_ok(cs_dtype)) return ASVD_E_BADARG;
    Plan p;
    int rc = make_plan(batch, m, n, 0, 0, p);
    if (rc) return rc;
    if (k < 1 || k > p.cols) return ASVD_E_BADARG;
    for (int b = 0; b < batch; ++b)
        if (!a_host[b] || !S_host[b]) return ASVD_E_BADARG;
    if (g_prof_enabled) prof_begin();
    rc = -100;
    if (tall_wanted(p))
        rc = svd_tall(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes,
                      info_host, stream);
    if (rc == -100)
        rc = svd_direct(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes,
                        info_host, stream, false);
    if (g_prof_enabled) prof_end();
    return rc;
}

int asvd_svd(const void* a, int a_dtype, int64_t m, int64_t n, int64_t lda, const void* col_scale, int cs_dtype, float* U,
             float* S, float* V, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes, int* info_host,
             void* stream) {
    const void* ap[1] = {a};
    const void* cp[1] = {col_scale};
    float* up[1] = {U};
    float* sp[1] = {S};
    float* vp[1] = {V};
    return asvd_svd_batched(1, ap, a_dtype, m, n, lda, col_scale ? cp : nullptr, cs_dtype, (U ? up : nullptr), sp,
                            (V ? vp : nullptr), k, max_sweeps, tol, work, work_bytes, info_host, stream);
}

}  // extern "C"
