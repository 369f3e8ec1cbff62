// jacobi_kernels.h — device kernels of the block-Jacobi sweeps (included by svd_jacobi.hip inside its anonymous namespace): pack, the
// single-level Gram / update pass, the two-level kernels (twolevel.h), the coupling snapshot of the sparse sweeps, and the finalize kernels
// (backsolve, column norms, ranking, gather).  The 64x64 eigen-solves live in evd_wave.hip; the host driver in svd_jacobi.hip.
#pragma once

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
// The termination measure is the snapshot's (same definition as in the eigen-solve, evd_wave.hip), so the stopping rule is unchanged.
__global__ __launch_bounds__(256) void panel_sumsq_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride,
                                                          int m_pad, int n_pad, float* __restrict__ dn, const int* __restrict__ done) {
    const int I = blockIdx.x, b = blockIdx.y, c = threadIdx.x & 31, g = threadIdx.x >> 5;
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
}

__device__ __forceinline__ float nanmax(float a, float b) { return (b != b) ? b : ((a != a) ? a : fmaxf(a, b)); }

// grid (ceil(nb/4), ceil(nb/4), batch); upper-triangular tiles only.  Wave w owns panel J = 4*jg + w against panels I = 4*ig + a.
// 32-row chunks of the eight panels are staged in LDS (coalesced 16-B loads, next chunk prefetched into registers while the
// current one is in the matrix pipe); every wave reads its operands from LDS as conflict-free 256-B rows.
__global__ __launch_bounds__(256) void fullcheck_kernel(Sched sc, const float* __restrict__ X, int64_t panel_stride, int64_t batch_stride, int nb,
                                                        int m_pad, int n_pad, const float* __restrict__ dn, float tol, int kb,
                                                        unsigned char* __restrict__ pflag, unsigned* __restrict__ maxoff_bits,
                                                        const int* __restrict__ done) {
    // (An XCD-aware tile order — the workgroups of one XCD walking a contiguous run of tiles — was measured SLOWER: 66.8 vs 55.6 ms for
    // the three snapshots of 32 problems; the plain order stays.  Two chunks of loads in flight per thread: 184 instead of 152 VGPRs, two
    // instead of three workgroups per CU, 44.1 vs 38.7 ms.)
    const int ig = blockIdx.x, jg = blockIdx.y, b = blockIdx.z;
    if (ig > jg || ld_flag(done + b)) return;
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
    const float* __restrict__ dnb = dn + (int64_t)b * n_pad;
    // power-of-two column scale from the column's own squared norm (just measured by panel_sumsq_kernel): scaled columns have norm <= 2^SG_TARGET
    auto col_exp = [&](float d) { return min(max(sg_half_exp(d), -60), 60) - SG_TARGET; };
    {
        // split-fp16 with power-of-two column scales (twolevel.h, round 4): each fp32 operand = two fp16 numbers (22 bits), three products per fp32
        // product on the fp16 matrix pipe — a cosine is resolved to 2^-22 of the norm product, what a coupling test against tol = 1e-6 needs, at
        // half the matrix work of the six-product bf16 form of rounds 2-3.  The operand of k-step ks of a panel is: lane (column cc, group hh) holds
        // rows 16 ks + 8 hh + e, e = 0..7.  Every operand of a 32-row chunk is built ONCE per workgroup: thread t owns the slots (panel
        // (t >> 7) + 2 j, k-step (t >> 6) & 1, lane t & 63), j = 0..3, loads its eight values — one column, so ONE scale — straight from global
        // memory (a wave-load covers two 128-byte row segments), splits them and stores the two parts as ready operands; the waves then only
        // read 16-byte operands.
        __shared__ u32x4 oimg[8 * 2 * 2 * 64];
        const int sks = (tid >> 6) & 1, sl = tid & 63;
        const float* src[4];
        float mulc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = (tid >> 7) + 2 * j;
            const int pnl = (q < 4) ? ig * 4 + q : jg * 4 + q - 4;
            const int pc = pnl < nb ? pnl : nb - 1;
            src[j] = Xb + (int64_t)pc * panel_stride + (int64_t)(16 * sks + 8 * (sl >> 5)) * PB + (sl & 31);
            mulc[j] = ldexpf(1.0f, -col_exp(dnb[pc * PB + (sl & 31)]));
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
                u32x4 p1, p2;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned x, y;
                    split2s(pre[j][2 * e2], pre[j][2 * e2 + 1], mulc[j], mulc[j], x, y);
                    p1[e2] = x; p2[e2] = y;
                }
                u32x4* o = oimg + ((((tid >> 7) + 2 * j) * 2 + sks) * 2) * 64 + sl;
                o[0] = p1; o[64] = p2;
            }
            __syncthreads();
            if (r0 + 32 < m_pad) fetch(r0 + 32);
            // eight stages (k-step, A panel) of three MFMAs; the operands of stage i + 1 are read from LDS before the MFMAs of stage i are
            // issued (left to the compiler each ds_read sat in front of its consumer: 51 % of the wave cycles waiting to issue)
            struct Op2 { h16x8 p1, p2; };
            auto ld2 = [&](int slot, int ks) {
                const u32x4* o = oimg + ((slot * 2 + ks) * 2) * 64 + lane;
                Op2 r;
                r.p1 = __builtin_bit_cast(h16x8, o[0]); r.p2 = __builtin_bit_cast(h16x8, o[64]);
                return r;
            };
            Op2 Bc = ld2(4 + w, 0), Ac = ld2(0, 0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int a = st & 3;
                Op2 An = Ac, Bn = Bc;
                if (st + 1 < 8) An = ld2((st + 1) & 3, (st + 1) >> 2);
                if (st == 3) Bn = ld2(4 + w, 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac.p2, Bc.p1, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac.p1, Bc.p2, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac.p1, Bc.p1, acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                Ac = An;
                if (st == 3) Bc = Bn;
            }
        }
    }
    if (J >= nb) return;
    const int h = lane >> 5, c = lane & 31;
    const float dj = dnb[J * PB + c];
    const int ej = col_exp(dj);
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
            const float g = ldexpf(acc[a][reg], col_exp(di) + ej);   // back to the scale of the data
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

