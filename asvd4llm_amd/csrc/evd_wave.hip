// evd_wave.hip — WAVE-LOCAL 64x64 symmetric eigen-solver of the block-Jacobi SVD (its own translation unit: built with -fno-slp-vectorize,
// see jacobi_shared.h; the host driver in svd_jacobi.hip reaches the kernels through the launch_evdw* functions at the end).
//
// Round 1/2 solved every 64x64 Gram block with one 256-thread workgroup on an LDS image: 64 phases x (rotation -> shuffle -> 2x2 block
// updates -> __syncthreads), ~3600 cycles per phase for the four solves a CU holds — a dependency chain through LDS and the workgroup
// barrier, 23 % of a bench step ("weak" #2).  Here ONE WAVE owns a solve and nothing leaves its registers during the sweep:
//
//   lane c  = column c of G (and of the accumulated eigenvector matrix Q);   register r = row r      (g[64] + q[64] VGPRs)
//
// Same ordering as before (odd-even transposition with swap, Luk-Park): phase A pairs positions (2k, 2k+1), phase B pairs (2k+1, 2k+2)
// (positions 0 and 63 idle), the rotated columns / rows exchange places, 64 phases = every pair once.  In this layout
//   * a ROW rotation pairs two registers with static indices — plain VALU on all lanes, coefficients (c_k, s_k) broadcast from the lanes
//     that computed them with v_readlane (SGPR operands);
//   * a COLUMN rotation pairs two lanes — the partner's value arrives as a DPP operand (quad_perm [1,0,3,2] in phase A; wave_shl:1 /
//     wave_shr:1 in phase B), fused into the FMA that consumes it;
//   * the pivot (a, d, b) of a lane's pair: a, d from a per-lane vector of the DIAGONAL kept in closed form (d + t b, a - t b — the
//     diagonal entries inside the register image are never read), b = g[lane + 1] of the lower lane, collected for the next phase with
//     one select per register while the registers are being written.  The annihilated element is not zeroed (it stays at rounding level).
//   No LDS, no barrier, no s_waitcnt inside the sweep: ~520 VALU instructions per phase and wave.  tools/proto_evd_wave.py is the CPU
//   prototype of exactly this data flow.
//
// Kernels:  evdw0_kernel  — single-level solves (internal step d = 1 of a two-level sweep, sparse rounds, small problems): four waves =
//                           four independent pairs per workgroup;
//           evdw12_kernel — BOTH inner steps of a super-pair in one launch (two waves: sub-pairs (0,2) (1,3), then (0,3) (1,2)); the
//                           step-0 eigenvectors and diagonal blocks stay in LDS, the epilogue emits Qfin and the carried blocks.
//                           Template RINGM: which inner steps visit only the 1024 CROSS pairs of their two panels, on a ring of interleaved
//                           positions in 32 phases instead of 64 (evdw_sweep_ring; round 5).  Default from 2048 columns on: step 1 (its panels
//                           arrive with exactly diagonal Gram blocks — step 0 has just diagonalised them — so the visit gives up nothing).

#include "common.h"
#include "jacobi_shared.h"
#include <cstdlib>
#include <cstdio>

namespace {
using namespace asvdk;

constexpr int DPP_XOR1 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int DPP_SHL1 = 0x130;   // wave_shl:1 — lane i reads lane i + 1
constexpr int DPP_SHR1 = 0x138;   // wave_shr:1 — lane i reads lane i - 1

template <int CTRL>
__device__ __forceinline__ float dppf(float x) {  // invalid source lanes (shifts at the wave ends) read as 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rdlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ int rdlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

// partner value of phase B (lane 2k+1 <-> lane 2k+2): odd lanes take lane + 1, even lanes lane - 1 — a full wave_shl:1 move, then a
// wave_shr:1 move restricted to the even lanes (bank_mask 0x5, bank = lane & 3) over it.  The idle lanes 0 / 63 get a finite value of no
// consequence (their partner weight is 0).
__device__ __forceinline__ float dpp_partner_b(float x) {
    const int xi = __float_as_int(x);
    const int p1 = __builtin_amdgcn_update_dpp(0, xi, DPP_SHL1, 0xF, 0xF, true);           // every lane <- lane + 1 (lane 63: 0); no tied `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(p1, xi, DPP_SHR1, 0xF, 0x5, false));  // even lanes <- lane - 1 (lane 0 keeps p1)
}
// column update of phase B in three instructions:  z = own * y + cl * y[lane + 1] + cr * y[lane - 1]   (cl = 0 on even, cr = 0 on odd and
// idle lanes).  The compiler fuses a DPP move into v_mul_f32 but not into v_fmac_f32, so the last term is written by hand.  Hazard
// (VALU write of y -> DPP read of y needs 2 wait states, which hipcc does not insert inside an asm): the asm depends on z, z on the
// v_mul_f32_dpp of the same y, and that one the compiler itself keeps >= 2 wait states behind y's producer.
__device__ __forceinline__ float col_update_b(float y, float own, float cl, float cr) {
    float z = fmaf(own, y, cl * dppf<DPP_SHL1>(y));
    asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(z) : "v"(y), "v"(cr));
    return z;
}
// RING variants (cross-only visits, see evdw_sweep_ring): phase B also pairs positions 63 and 0, so the partners come by wave ROTATES and no lane idles
constexpr int DPP_ROL1 = 0x134;   // wave_rol:1 — lane i reads lane (i + 1) & 63
constexpr int DPP_ROR1 = 0x13C;   // wave_ror:1 — lane i reads lane (i - 1) & 63
__device__ __forceinline__ float col_update_ring(float y, float own, float cl, float cr) {
    float z = fmaf(own, y, cl * dppf<DPP_ROL1>(y));
    asm("v_fmac_f32_dpp %0, %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(z) : "v"(y), "v"(cr));
    return z;
}
// one element of the next phase's pivot vector: bn[L] = v[L].  `lane` is made opaque once per phase (an empty asm) so that the 32 lane
// masks of a phase are not hoisted out of the sweep loop and kept in (spilled) SGPRs.
__device__ __forceinline__ float put_lane(float bn, float v, int L, int lane) { return (lane == L) ? v : bn; }
// The same select with the lane mask built on the SCALAR unit (one s_lshl_b64 of an opaque 1) instead of a v_cmp: one VALU instruction per
// register instead of two (a compare + select pair costs 2.75 ns on a saturated SIMD, tools/ubench/valu_cost.hip), and nothing for the
// compiler to hoist out of the sweep loop as 32 live masks — `one` is redefined (empty asm) once per phase.
__device__ __forceinline__ float put_lane_s(float bn, float v, int L, unsigned long long one) {
    const unsigned long long m = one << L;
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(bn), "v"(v), "s"(m));
    return r;
}

// One phase.  bpiv: on entry the pivot b = G[lane + 1][lane] in the LOWER lane of every pair of this phase; on exit the same for the
// next phase (the other parity).  diag: G[lane][lane], closed form.
// Row coefficients: the (c, s) of row pair k are those of the lanes at the same positions.  They reach all lanes through 512 bytes of
// wave-private LDS (one ds_write_b64 per lane, one broadcast ds_read_b64 per row pair) instead of two v_readlane per row pair: measured
// 2.3 ns per v_readlane against 1.1 ns per plain VALU instruction on a saturated SIMD (tools/ubench/valu_cost.hip) — 14 % of a phase —
// while the LDS pipe is idle during the sweep.  A wave's LDS operations complete in order: no barrier.
template <int PAR, bool RING = false>
__device__ __forceinline__ void evdw_phase(float (&g)[64], float (&q)[64], float& diag, float& bpiv, const int lane, float* __restrict__ cslds) {
    const bool odd = (lane & 1) != 0;
    float bn = 0.0f;
    unsigned long long one = 1ull;
    asm volatile("" : "+s"(one));
    if constexpr (PAR == 0) {
        const bool lower = !odd;
        const float dpart = dppf<DPP_XOR1>(diag), bo = dppf<DPP_XOR1>(bpiv);
        const float b = lower ? bpiv : bo;
        const float a_ = lower ? diag : dpart, d_ = lower ? dpart : diag;
        float c, s, t;
        jacobi_rot(a_, d_, b, c, s, t);
        diag = fmaf(lower ? t : -t, b, lower ? d_ : a_);   // position p now holds the rotated q and vice versa (swap): d + t b | a - t b
        const float own = lower ? s : -s;                   // new = own * x + c * x_partner
        *(float2*)(cslds + 2 * lane) = make_float2(c, s);
        // row coefficients in batches of eight broadcast reads, the next batch issued before the current one is consumed: left to the
        // compiler every ds_read_b64 sat right in front of its row pair and the LDS latency was exposed some twenty times per phase
        float2 csb[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) csb[0][j] = *(const float2*)(cslds + 2 * (2 * j));
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if ((k & 7) == 0) {
                if (k + 8 < 32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) csb[((k >> 3) + 1) & 1][j] = *(const float2*)(cslds + 2 * (2 * (k + 8 + j)));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const float2 cs2 = csb[(k >> 3) & 1][k & 7];
            const float ck = cs2.x, sk = cs2.y;
            const float x0 = g[2 * k], x1 = g[2 * k + 1];
            const float y0 = fmaf(ck, x1, sk * x0);         // rows: new[p] = s row[p] + c row[q] ; new[q] = c row[p] - s row[q]
            const float y1 = fmaf(-sk, x1, ck * x0);
            const float z0 = fmaf(own, y0, c * dppf<DPP_XOR1>(y0));   // v_mul_f32_dpp + v_fmac
            const float z1 = fmaf(own, y1, c * dppf<DPP_XOR1>(y1));
            g[2 * k] = z0;
            g[2 * k + 1] = z1;
            if (k >= 1) bn = put_lane_s(bn, z0, 2 * k - 1, one);    // phase B: lower lanes are odd L, their pivot is register L + 1
            else if (RING) bn = put_lane_s(bn, z0, 63, one);        // ring: position 63 pairs with position 0, its pivot is register 0
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) q[r] = fmaf(own, q[r], c * dppf<DPP_XOR1>(q[r]));
    } else if constexpr (RING) {
        // phase B on the ring: pairs (2k + 1, (2k + 2) & 63), k = 0..31 — nothing idles
        const bool lower = odd;
        const float d_up = dppf<DPP_ROL1>(diag), d_dn = dppf<DPP_ROR1>(diag), b_dn = dppf<DPP_ROR1>(bpiv);
        const float b = lower ? bpiv : b_dn;
        const float a_ = lower ? diag : d_dn, d_ = lower ? d_up : diag;
        float c, s, t;
        jacobi_rot(a_, d_, b, c, s, t);
        diag = fmaf(lower ? t : -t, b, lower ? d_ : a_);
        const float own = lower ? s : -s;
        const float cl = lower ? c : 0.0f;   // weight of the value of lane + 1 (odd lanes)
        const float cr = lower ? 0.0f : c;   // weight of the value of lane - 1 (even lanes)
        *(float2*)(cslds + 2 * lane) = make_float2(c, s);
        float2 csb[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) csb[0][j] = *(const float2*)(cslds + 2 * (2 * j + 1));
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if ((k & 7) == 0) {
                if (k + 8 < 32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) csb[((k >> 3) + 1) & 1][j] = *(const float2*)(cslds + 2 * (2 * (k + 8 + j) + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const float2 cs2 = csb[(k >> 3) & 1][k & 7];
            const float ck = cs2.x, sk = cs2.y;
            const int rp = 2 * k + 1, rq = (2 * k + 2) & 63;
            const float x0 = g[rp], x1 = g[rq];
            const float y0 = fmaf(ck, x1, sk * x0);
            const float y1 = fmaf(-sk, x1, ck * x0);
            const float z0 = col_update_ring(y0, own, cl, cr);
            const float z1 = col_update_ring(y1, own, cl, cr);
            g[rp] = z0;
            g[rq] = z1;
            bn = put_lane_s(bn, z0, 2 * k, one);                    // phase A: lower lanes are even L, their pivot is register L + 1
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) q[r] = col_update_ring(q[r], own, cl, cr);
    } else {
        const bool lower = odd;                              // pairs (2k+1, 2k+2); lanes 0 and 63 idle
        const bool idle = (lane == 0) || (lane == 63);
        const float d_up = dppf<DPP_SHL1>(diag), d_dn = dppf<DPP_SHR1>(diag), b_dn = dppf<DPP_SHR1>(bpiv);
        const float b = lower ? bpiv : b_dn;
        const float a_ = lower ? diag : d_dn, d_ = lower ? d_up : diag;
        float c, s, t;
        jacobi_rot(a_, d_, b, c, s, t);
        const float nd = fmaf(lower ? t : -t, b, lower ? d_ : a_);
        diag = idle ? diag : nd;
        const float own = idle ? 1.0f : (lower ? s : -s);
        const float cl = (lower && !idle) ? c : 0.0f;        // weight of the value of lane + 1 (odd lanes)
        const float cr = (!lower && !idle) ? c : 0.0f;       // weight of the value of lane - 1 (even lanes)
        c = idle ? 1.0f : c;                                 // rows 0 and 63 are idle too; the row loop below never reads lanes 0 / 63
        s = idle ? 0.0f : s;
        *(float2*)(cslds + 2 * lane) = make_float2(c, s);
        g[0] = col_update_b(g[0], own, cl, cr);    // row 0 is idle: columns only
        float2 csb[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) csb[0][j] = *(const float2*)(cslds + 2 * (2 * j + 1));
#pragma unroll
        for (int k = 0; k < 31; ++k) {
            if ((k & 7) == 0) {
                if (k + 8 < 31) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k + 8 + j < 31) csb[((k >> 3) + 1) & 1][j] = *(const float2*)(cslds + 2 * (2 * (k + 8 + j) + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const float2 cs2 = csb[(k >> 3) & 1][k & 7];
            const float ck = cs2.x, sk = cs2.y;
            const float x0 = g[2 * k + 1], x1 = g[2 * k + 2];
            const float y0 = fmaf(ck, x1, sk * x0);
            const float y1 = fmaf(-sk, x1, ck * x0);
            const float z0 = col_update_b(y0, own, cl, cr);
            const float z1 = col_update_b(y1, own, cl, cr);
            g[2 * k + 1] = z0;
            g[2 * k + 2] = z1;
            bn = put_lane_s(bn, z0, 2 * k, one);                    // phase A: lower lanes are even L, their pivot is register L + 1
        }
        {   // row 63 is idle
            const float z = col_update_b(g[63], own, cl, cr);
            g[63] = z;
            bn = put_lane_s(bn, z, 62, one);
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) q[r] = col_update_b(q[r], own, cl, cr);
    }
    bpiv = bn;
}

// diagonal / first pivots of a freshly loaded image:  diag = g[lane] ;  bpiv = g[lane + 1]  (dynamic register index -> select chains, once)
__device__ __forceinline__ void evdw_init_state(const float (&g)[64], const int lane, float& diag, float& bpiv) {
    float d = 0.0f, b = 0.0f;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        d = (lane == r) ? g[r] : d;
        if (r >= 1) b = (lane == r - 1) ? g[r] : b;
    }
    diag = d;
    bpiv = b;
}

// Scaled off-diagonal measures of the image (definitions of evd_body): off0 = max |g_ij| / sqrt(g_ii g_jj) decides whether the pair
// rotates; offt = max |g_ij| / max(g_ii, g_jj) over entries that touch a LEADING panel is what termination looks at.  NaN / Inf anywhere
// in the image -> both NaN.  top_lo / top_hi: positions 0..31 / 32..63 belong to a leading panel.  Wave-uniform results.
__device__ __forceinline__ void evdw_measure(const float (&g)[64], const float diag, const int lane, const bool top_lo, const bool top_hi,
                                            float& off0, float& offt) {
    const float rs_c = diag > 0.0f ? __builtin_amdgcn_rsqf(diag) : 0.0f;
    const bool lead_c = lane < 32 ? top_lo : top_hi;
    float loc = 0.0f, loct = 0.0f, poison = 0.0f;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const float dr = rdlane(diag, r), rs_r = rdlane(rs_c, r);
        const float ag = fabsf(g[r]);
        poison = fmaf(g[r], 0.0f, poison);                   // NaN / Inf -> NaN, finite -> unchanged
        const bool self = lane == r;
        loc = fmaxf(loc, self ? 0.0f : ag * rs_r);
        const bool lead = lead_c || (r < 32 ? top_lo : top_hi);
        const float mx = fmaxf(dr, diag);
        const float vt = mx > 0.0f ? ag * __builtin_amdgcn_rcpf(mx) : 0.0f;
        loct = fmaxf(loct, (self || !lead) ? 0.0f : vt);
    }
    loc *= rs_c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        loc = fmaxf(loc, __shfl_xor(loc, o, 64));
        loct = fmaxf(loct, __shfl_xor(loct, o, 64));
        poison += __shfl_xor(poison, o, 64);
    }
    const bool bad = poison != poison;
    off0 = bad ? __builtin_nanf("") : loc;
    offt = bad ? __builtin_nanf("") : loct;
}

// the sweeps: `loops` full odd-even cycles of 64 phases each are  loops * 32  (A, B) phase pairs
constexpr int EVDW_CS_FLOATS = 128;   // wave-private LDS of the row coefficients
__device__ __forceinline__ void evdw_sweep(float (&g)[64], float (&q)[64], float& diag, float& bpiv, const int lane, const int phase_pairs,
                                           float* __restrict__ cslds) {
#pragma unroll 1
    for (int ph2 = 0; ph2 < phase_pairs; ++ph2) {
        evdw_phase<0>(g, q, diag, bpiv, lane, cslds);
        evdw_phase<1>(g, q, diag, bpiv, lane, cslds);
    }
}

__device__ __forceinline__ void evdw_identity(float (&q)[64], const int lane) {
#pragma unroll
    for (int r = 0; r < 64; ++r) q[r] = (lane == r) ? 1.0f : 0.0f;
}

// ---- cross-only visits on a ring (round 5) ----------------------------------------------------------------------------------------------
// A solve of the two-level sweeps pairs two 32-column panels S, T whose own Gram blocks arrive (nearly) diagonal: what the visit has to
// annihilate are the 1024 CROSS couplings; the 2 x 496 pairs inside the panels belong to the internal step of the sweep.  With the positions
// INTERLEAVED (position 2k = column k of S, 2k + 1 = column k of T) and phase B closed to a ring (position 63 pairs with position 0), the
// odd-even transposition with swap moves the S columns one way round and the T columns the other: after 32 phases — half a full cycle —
// every S column has met every T column exactly once and no two columns of the same panel have met (tools/proto_cross_only.py: one more
// sparse sweep at the end of the outer iteration, half the phases per visit).
//   ring position of natural index x:  pos(x) = 2 x (x < 32),  2 (x - 32) + 1 (x >= 32);   natural index of position p:  nat(p) = (p & 1) * 32 + (p >> 1)
__device__ __forceinline__ constexpr int ring_pos(int x) { return x < 32 ? 2 * x : 2 * (x - 32) + 1; }
// image natural -> ring, rows (registers, a renaming) and columns (lanes: one ds_bpermute per register)
__device__ __forceinline__ void ring_image(float (&g)[64], const int lane) {
    const int src = (((lane & 1) << 5) + (lane >> 1)) << 2;   // byte address of the lane that holds natural column nat(lane)
    float t[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) t[ring_pos(r)] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(g[r])));
#pragma unroll
    for (int r = 0; r < 64; ++r) g[r] = t[r];
}
// rows of Q back to natural order (its columns = positions stay: every consumer places them by their sorted rank)
__device__ __forceinline__ void ring_rows_to_natural(float (&q)[64]) {
    float t[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) t[r] = q[ring_pos(r)];
#pragma unroll
    for (int r = 0; r < 64; ++r) q[r] = t[r];
}
// `loops` cross-only visits (16 phase pairs each).  In: the natural image in g, q = identity.  Out: g = the image in RING positions (rows and
// lanes: what evdw_store_diag_blocks / evdw_finish address by position), q = eigenvector matrix with natural rows, diag per position.
__device__ __forceinline__ void evdw_sweep_ring(float (&g)[64], float (&q)[64], float& diag, float& bpiv, const int lane, const int loops,
                                                float* __restrict__ cslds) {
    ring_image(g, lane);
    evdw_init_state(g, lane, diag, bpiv);
#pragma unroll 1
    for (int ph2 = 0; ph2 < 16 * loops; ++ph2) {
        evdw_phase<0, true>(g, q, diag, bpiv, lane, cslds);
        evdw_phase<1, true>(g, q, diag, bpiv, lane, cslds);
    }
    ring_rows_to_natural(q);
}

// After the sweeps: cs = 1 / |q_c| (fp64 norm: the accumulated rounding of ~64 rotations per column must not drift the norms of the
// updated panels) and rnk = position of column `lane` in the descending order of the eigenvalues (ties by index).  A solve that did not
// rotate keeps everything in place (cs = 1, rnk = lane).
__device__ __forceinline__ void evdw_finish(const float (&q)[64], const float diag, const int lane, const bool rotate, float& cs, int& rnk) {
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const double v = (double)q[r];
        acc = fma(v, v, acc);
    }
    cs = (rotate && acc > 0.0) ? (float)(1.0 / sqrt(acc)) : 1.0f;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const float o = rdlane(diag, i);
        cnt += (o > diag || (o == diag && i < lane)) ? 1 : 0;
    }
    rnk = rotate ? cnt : lane;
}

// transformed diagonal 32x32 blocks of the sorted, rescaled matrix: block h = sorted positions 32h .. 32h+31.  The diagonal itself comes
// from the closed-form vector.  d0 / d1 may be global or LDS pointers; every store writes (part of) one 128-byte row.
__device__ __forceinline__ void evdw_store_diag_blocks(const float (&g)[64], const float diag, const float cs, const int rnk, const int lane,
                                                       float* __restrict__ d0, float* __restrict__ d1) {
    const int myh = rnk >> 5, myc = rnk & 31;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const int rr = rdlane_i(rnk, r);
        const float csr = rdlane(cs, r);
        const float v = ((lane == r) ? diag : g[r]) * csr * cs;
        float* dst = (rr >> 5) ? d1 : d0;
        if ((rr >> 5) == myh) dst[(rr & 31) * 32 + myc] = v;
    }
}

// --------------------------------------------------------------------------------------------------
// Cooperative sweep (latency form): NW waves share ONE solve, wave h holds rows r0 = h * 64 / NW .. of the image and of Q (lane = column
// as before).  A lone wave issues one VALU instruction per ~4.7 cycles, so a launch with fewer solves than SIMDs (batch 1, 768-column
// problems, any batch <= 8 at 4096 columns) waits 91 us per sweep with 3/4 of the chip idle; split four ways a phase is ~180 instead of
// ~600 instructions per wave plus one workgroup barrier.  Same arithmetic, element for element, as evdw_phase (bit-identical results):
//   * every wave computes all 64 rotations itself (diagonal vector and pivot vector are replicated) and rotates only its own rows;
//   * the pivots of the next phase are elements of rows that live in different waves: each wave publishes the ones it owns into a
//     shared vector (one per parity: written in one phase, read in the next, rewritten two barriers later);
//   * phase B pairs rows (2k+1, 2k+2): the pair at a block boundary needs the neighbour's boundary row as it was after phase A —
//     published together with the pivots, one barrier per phase in total.
// LDS per solve (floats): XG 64x64 (scatter / gather of the image) | XQ 64x64 (gather of Q) | PIV 2x64 | BROW 2 x NW x 64 | CS NW x 128 | DIAG 64 | FLAG 4
template <int NW> struct Coop {
    static constexpr int RB = 64 / NW;
    static constexpr int XG = 0, XQ = 4096, PIV = 8192, BROW = PIV + 128, CS = BROW + 2 * NW * 64, DIAG = CS + NW * 128, FLAG = DIAG + 64;
    static constexpr int FLOATS = FLAG + 4;
};

template <int NW, bool RING = false>
__device__ __forceinline__ void coop_phase_a(float (&gl)[64 / NW], float (&ql)[64 / NW], float& diag, const float bpiv, const int lane, const int r0,
                                             const int h, float* __restrict__ L) {
    constexpr int RB = 64 / NW;
    float* __restrict__ cslds = L + Coop<NW>::CS + h * 128;
    const bool odd = (lane & 1) != 0, lower = !odd;
    const float dpart = dppf<DPP_XOR1>(diag), bo = dppf<DPP_XOR1>(bpiv);
    const float b = lower ? bpiv : bo;
    const float a_ = lower ? diag : dpart, d_ = lower ? dpart : diag;
    float c, s, t;
    jacobi_rot(a_, d_, b, c, s, t);
    diag = fmaf(lower ? t : -t, b, lower ? d_ : a_);
    const float own = lower ? s : -s;
    *(float2*)(cslds + 2 * lane) = make_float2(c, s);
    float bn = 0.0f;
    unsigned long long one = 1ull;
    asm volatile("" : "+s"(one));
    float2 csb[RB / 2];   // all row coefficients of this wave in one batch of broadcast reads (see evdw_phase)
#pragma unroll
    for (int kp = 0; kp < RB / 2; ++kp) csb[kp] = *(const float2*)(cslds + 2 * (r0 + 2 * kp));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kp = 0; kp < RB / 2; ++kp) {
        const float2 cs2 = csb[kp];
        const float ck = cs2.x, sk = cs2.y;
        const float x0 = gl[2 * kp], x1 = gl[2 * kp + 1];
        const float y0 = fmaf(ck, x1, sk * x0);
        const float y1 = fmaf(-sk, x1, ck * x0);
        const float z0 = fmaf(own, y0, c * dppf<DPP_XOR1>(y0));
        const float z1 = fmaf(own, y1, c * dppf<DPP_XOR1>(y1));
        gl[2 * kp] = z0;
        gl[2 * kp + 1] = z1;
        if (kp > 0 || h > 0) bn = put_lane_s(bn, z0, r0 + 2 * kp - 1, one);   // phase B: lower lanes are odd L, their pivot is row L + 1 (row 0 has none)
        else if (RING) bn = put_lane_s(bn, z0, 63, one);                       // ring: position 63 pairs with position 0
    }
    if (odd && ((lane + 1) & (RING ? 63 : 127)) >= r0 && ((lane + 1) & (RING ? 63 : 127)) < r0 + RB) L[Coop<NW>::PIV + lane] = bn;   // pivots of phase B (parity buffer 0)
    L[Coop<NW>::BROW + (2 * h) * 64 + lane] = gl[0];                                           // boundary rows as phase B will find them
    L[Coop<NW>::BROW + (2 * h + 1) * 64 + lane] = gl[RB - 1];
#pragma unroll
    for (int i = 0; i < RB; ++i) ql[i] = fmaf(own, ql[i], c * dppf<DPP_XOR1>(ql[i]));        // Q last: the stores above are under way meanwhile
}

template <int NW>
__device__ __forceinline__ void coop_phase_b(float (&gl)[64 / NW], float (&ql)[64 / NW], float& diag, const float bpiv, const int lane, const int r0,
                                             const int h, float* __restrict__ L) {
    constexpr int RB = 64 / NW;
    float* __restrict__ cslds = L + Coop<NW>::CS + h * 128;
    const bool odd = (lane & 1) != 0, lower = odd;
    const bool idle = (lane == 0) || (lane == 63);
    const float d_up = dppf<DPP_SHL1>(diag), d_dn = dppf<DPP_SHR1>(diag), b_dn = dppf<DPP_SHR1>(bpiv);
    const float b = lower ? bpiv : b_dn;
    const float a_ = lower ? diag : d_dn, d_ = lower ? d_up : diag;
    float c, s, t;
    jacobi_rot(a_, d_, b, c, s, t);
    const float nd = fmaf(lower ? t : -t, b, lower ? d_ : a_);
    diag = idle ? diag : nd;
    const float own = idle ? 1.0f : (lower ? s : -s);
    const float cl = (lower && !idle) ? c : 0.0f;
    const float cr = (!lower && !idle) ? c : 0.0f;
    c = idle ? 1.0f : c;
    s = idle ? 0.0f : s;
    *(float2*)(cslds + 2 * lane) = make_float2(c, s);
    float bn = 0.0f;
    unsigned long long one = 1ull;
    asm volatile("" : "+s"(one));
    // every LDS value of the phase in one batch: coefficients of the pair below the block, of the interior pairs and of the pair above, and
    // the two neighbour rows (the idle ends read a valid address and ignore the value)
    float2 csb[RB / 2 + 1];
    csb[0] = *(const float2*)(cslds + 2 * (h == 0 ? 0 : r0 - 1));
#pragma unroll
    for (int j = 0; j < RB / 2 - 1; ++j) csb[1 + j] = *(const float2*)(cslds + 2 * (r0 + 2 * j + 1));
    csb[RB / 2] = *(const float2*)(cslds + 2 * (r0 + RB - 1));
    const float xp_n = L[Coop<NW>::BROW + (2 * (h == 0 ? 0 : h - 1) + 1) * 64 + lane];
    const float xq_n = L[Coop<NW>::BROW + (2 * (h == NW - 1 ? h : h + 1)) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    if (h == 0) {   // row 0 is idle: columns only
        gl[0] = col_update_b(gl[0], own, cl, cr);
    } else {        // upper member of the pair (r0 - 1, r0): the lower member is the previous wave's last row
        const float2 cs2 = csb[0];
        gl[0] = col_update_b(fmaf(-cs2.y, gl[0], cs2.x * xp_n), own, cl, cr);
    }
#pragma unroll
    for (int j = 0; j < RB / 2 - 1; ++j) {
        const int pl = 2 * j + 1;
        const float2 cs2 = csb[1 + j];
        const float ck = cs2.x, sk = cs2.y;
        const float x0 = gl[pl], x1 = gl[pl + 1];
        const float y0 = fmaf(ck, x1, sk * x0);
        const float y1 = fmaf(-sk, x1, ck * x0);
        const float z0 = col_update_b(y0, own, cl, cr);
        const float z1 = col_update_b(y1, own, cl, cr);
        gl[pl] = z0;
        gl[pl + 1] = z1;
        bn = put_lane_s(bn, z0, r0 + pl - 1, one);  // phase A: lower lanes are even L, their pivot is row L + 1
    }
    if (h == NW - 1) {   // row 63 is idle
        const float z = col_update_b(gl[RB - 1], own, cl, cr);
        gl[RB - 1] = z;
        bn = put_lane_s(bn, z, 62, one);
    } else {             // lower member of the pair (r0 + RB - 1, r0 + RB): the upper member is the next wave's first row
        const float2 cs2 = csb[RB / 2];
        const float z0 = col_update_b(fmaf(cs2.x, xq_n, cs2.y * gl[RB - 1]), own, cl, cr);
        gl[RB - 1] = z0;
        bn = put_lane_s(bn, z0, r0 + RB - 2, one);
    }
    if (!odd && lane + 1 >= r0 && lane + 1 < r0 + RB) L[Coop<NW>::PIV + 64 + lane] = bn;      // pivots of phase A (parity buffer 1)
#pragma unroll
    for (int i = 0; i < RB; ++i) ql[i] = col_update_b(ql[i], own, cl, cr);
}

// phase B on the ring (cross-only visits): every row pair (2k + 1, (2k + 2) & 63) exists, the pair at EVERY block boundary — the one between the
// last and the first wave included — takes its other member from the neighbour's published boundary row
template <int NW>
__device__ __forceinline__ void coop_phase_b_ring(float (&gl)[64 / NW], float (&ql)[64 / NW], float& diag, const float bpiv, const int lane, const int r0,
                                                  const int h, float* __restrict__ L) {
    constexpr int RB = 64 / NW;
    float* __restrict__ cslds = L + Coop<NW>::CS + h * 128;
    const bool lower = (lane & 1) != 0;
    const float d_up = dppf<DPP_ROL1>(diag), d_dn = dppf<DPP_ROR1>(diag), b_dn = dppf<DPP_ROR1>(bpiv);
    const float b = lower ? bpiv : b_dn;
    const float a_ = lower ? diag : d_dn, d_ = lower ? d_up : diag;
    float c, s, t;
    jacobi_rot(a_, d_, b, c, s, t);
    diag = fmaf(lower ? t : -t, b, lower ? d_ : a_);
    const float own = lower ? s : -s;
    const float cl = lower ? c : 0.0f;
    const float cr = lower ? 0.0f : c;
    *(float2*)(cslds + 2 * lane) = make_float2(c, s);
    float bn = 0.0f;
    unsigned long long one = 1ull;
    asm volatile("" : "+s"(one));
    float2 csb[RB / 2 + 1];
    csb[0] = *(const float2*)(cslds + 2 * ((r0 + 63) & 63));                 // pair (r0 - 1, r0): coefficients at position r0 - 1
#pragma unroll
    for (int j = 0; j < RB / 2 - 1; ++j) csb[1 + j] = *(const float2*)(cslds + 2 * (r0 + 2 * j + 1));
    csb[RB / 2] = *(const float2*)(cslds + 2 * (r0 + RB - 1));                // pair (r0 + RB - 1, r0 + RB)
    const float xp_n = L[Coop<NW>::BROW + (2 * ((h + NW - 1) % NW) + 1) * 64 + lane];   // last row of the previous wave (as phase A left it)
    const float xq_n = L[Coop<NW>::BROW + (2 * ((h + 1) % NW)) * 64 + lane];            // first row of the next wave
    __builtin_amdgcn_sched_barrier(0);
    {   // upper member of the pair (r0 - 1, r0)
        const float2 cs2 = csb[0];
        gl[0] = col_update_ring(fmaf(-cs2.y, gl[0], cs2.x * xp_n), own, cl, cr);
    }
#pragma unroll
    for (int j = 0; j < RB / 2 - 1; ++j) {
        const int pl = 2 * j + 1;
        const float2 cs2 = csb[1 + j];
        const float ck = cs2.x, sk = cs2.y;
        const float x0 = gl[pl], x1 = gl[pl + 1];
        const float y0 = fmaf(ck, x1, sk * x0);
        const float y1 = fmaf(-sk, x1, ck * x0);
        const float z0 = col_update_ring(y0, own, cl, cr);
        const float z1 = col_update_ring(y1, own, cl, cr);
        gl[pl] = z0;
        gl[pl + 1] = z1;
        bn = put_lane_s(bn, z0, r0 + pl - 1, one);
    }
    {   // lower member of the pair (r0 + RB - 1, r0 + RB)
        const float2 cs2 = csb[RB / 2];
        const float z0 = col_update_ring(fmaf(cs2.x, xq_n, cs2.y * gl[RB - 1]), own, cl, cr);
        gl[RB - 1] = z0;
        bn = put_lane_s(bn, z0, r0 + RB - 2, one);
    }
    if (!lower && lane + 1 >= r0 && lane + 1 < r0 + RB) L[Coop<NW>::PIV + 64 + lane] = bn;      // pivots of phase A (parity buffer 1)
#pragma unroll
    for (int i = 0; i < RB; ++i) ql[i] = col_update_ring(ql[i], own, cl, cr);
}

// One full inner sweep (32 phase pairs) of the solve whose LDS block is L.  MAIN (h == 0) enters with the image in g, the state in
// diag / bpiv and `rotate`; it leaves with g, q, diag as evdw_sweep would (q = identity when the solve does not rotate).  Helpers pass
// dummies.  Every wave of the workgroup executes the same 66 barriers whether its solve rotates or not.
// RING: the cross-only visit (evdw_sweep_ring): the scatter of the image puts rows AND columns at their ring positions, 16 phase pairs, the gather
// returns the image in ring positions and Q with natural rows — element for element what the wave-local ring sweep computes.
template <int NW, bool RING = false>
__device__ __forceinline__ void coop_sweep(float (&g)[64], float (&q)[64], float& diag, const float bpiv_in, const bool rotate, const int lane, const int h,
                                           float* __restrict__ L) {
    constexpr int RB = 64 / NW;
    const int r0 = h * RB;
    const int plane = RING ? (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1) : lane;   // ring position of this lane's natural column
    if (h == 0) {
        if (rotate) {
#pragma unroll
            for (int r = 0; r < 64; ++r) L[Coop<NW>::XG + (RING ? ring_pos(r) : r) * 64 + plane] = g[r];
        }
        L[Coop<NW>::DIAG + plane] = diag;
        L[Coop<NW>::PIV + 64 + lane] = bpiv_in;   // first phase is A: parity buffer 1 (RING: replaced below, the natural pivots mean nothing there)
        L[Coop<NW>::PIV + lane] = 0.0f;
        if (lane == 0) ((int*)L)[Coop<NW>::FLAG] = rotate ? 1 : 0;
    }
    __syncthreads();
    const bool live = ((const int*)L)[Coop<NW>::FLAG] != 0;
    float gl[RB], ql[RB];
    float dg = L[Coop<NW>::DIAG + lane];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        gl[i] = live ? L[Coop<NW>::XG + (r0 + i) * 64 + lane] : 0.0f;
        ql[i] = (lane == r0 + i) ? 1.0f : 0.0f;
    }
    // RING: first pivots in ring positions, G'[lane + 1][lane] (only the even lanes' values are used); lane 63 gets 0 like evdw_init_state
    const float bp0 = (RING && live && lane < 63) ? L[Coop<NW>::XG + (lane + 1) * 64 + lane] : 0.0f;
#pragma unroll 1
    for (int ph2 = 0; ph2 < (RING ? 16 : 32); ++ph2) {
        if (live) coop_phase_a<NW, RING>(gl, ql, dg, (RING && ph2 == 0) ? bp0 : L[Coop<NW>::PIV + 64 + lane], lane, r0, h, L);
        __syncthreads();
        if (live) {
            if constexpr (RING) coop_phase_b_ring<NW>(gl, ql, dg, L[Coop<NW>::PIV + lane], lane, r0, h, L);
            else coop_phase_b<NW>(gl, ql, dg, L[Coop<NW>::PIV + lane], lane, r0, h, L);
        }
        __syncthreads();
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            L[Coop<NW>::XG + (r0 + i) * 64 + lane] = gl[i];
            L[Coop<NW>::XQ + (r0 + i) * 64 + lane] = ql[i];
        }
    }
    __syncthreads();
    if (h == 0 && live) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            g[r] = L[Coop<NW>::XG + r * 64 + lane];
            q[r] = L[Coop<NW>::XQ + (RING ? ring_pos(r) : r) * 64 + lane];
        }
        diag = dg;
    }
}

// --------------------------------------------------------------------------------------------------
// Single-level solves: the body of evd_kernel<0, KEEPG>, one WAVE per pair, four pairs per 256-thread workgroup.
// LDS: one 32 x 33 transposer per wave (the JI block of the image is the mirror of the stored IJ block).
constexpr int EVDW_TR_FLOATS = 32 * 33;

// NW = 1: four independent pairs per workgroup.  NW = 4 (latency form, launches with few pairs): ONE pair per workgroup, wave 0 is the
// main wave, waves 1..3 only help in the sweep (coop_sweep); every exit below is uniform over the workgroup.
template <int KEEPG, int NW>
__global__ __launch_bounds__(256, 2) void evdw0_kernel(Sched sc, const float* __restrict__ Gpart, int nsplit, float* __restrict__ Qbuf,
                                                       int* __restrict__ active, unsigned* __restrict__ maxoff_bits, int* __restrict__ nrot,
                                                       const int* __restrict__ done, float tol, int inner_sweeps, int nb, int step, int kb,
                                                       const int* __restrict__ plist, int list_stride, int npairs, EvdV3 v3) {
    __shared__ float trbuf[NW == 1 ? 4 : 1][EVDW_TR_FLOATS];
    __shared__ __attribute__((aligned(16))) float csbuf[NW == 1 ? 4 : 1][EVDW_CS_FLOATS];
    __shared__ __attribute__((aligned(16))) float coopbuf[NW == 1 ? 4 : Coop<NW == 1 ? 4 : NW>::FLOATS];
    const int lane = threadIdx.x & 63, wv_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wv = NW == 1 ? wv_ : 0, hw = NW == 1 ? 0 : wv_;
    const int pair = NW == 1 ? blockIdx.x * 4 + wv_ : blockIdx.x, b = blockIdx.y;
    if (pair >= npairs || ld_flag(done + b)) return;   // NW = 1: no workgroup barrier below, the waves are independent
    const int64_t slot = (int64_t)b * npairs + pair;
    int* act_flag = active + slot;
    int I, J;
    if (!get_pair(sc, plist, list_stride, b, nb, step, pair, I, J)) {  // padding pair / empty slot: nothing to rotate
        if (lane == 0 && hw == 0) *act_flag = 0;
        return;
    }
    float g[64], q[64];
    if constexpr (NW > 1) {
        if (hw > 0) {
            float dd = 0.0f;
            coop_sweep<NW>(g, q, dd, 0.0f, false, lane, hw, coopbuf);
            return;
        }
    }
    {
        // image rows 0..31: [II | IJ] read as two 128-byte row segments per register; rows 32..63: the JJ block for the upper lanes,
        // the lower lanes get IJ^T through the transposer.  Partials are summed in ascending order (as evd_body does).
        const float* __restrict__ gp = Gpart + slot * nsplit * 3072;
        const int hi = lane >> 5, cc = lane & 31;
#pragma unroll
        for (int r = 0; r < 64; ++r) g[r] = 0.0f;
#pragma unroll 2
        for (int s2 = 0; s2 < nsplit; ++s2) {
            const float* __restrict__ p = gp + (int64_t)s2 * 3072;
#pragma unroll
            for (int r = 0; r < 32; ++r) g[r] += p[hi * 1024 + r * 32 + cc];
#pragma unroll
            for (int r = 0; r < 32; ++r) g[32 + r] += p[2048 + r * 32 + cc];
        }
        float* tr = trbuf[wv];
        if (hi) {
#pragma unroll
            for (int r = 0; r < 32; ++r) tr[cc * 33 + r] = g[r];      // IJ[r][cc]  ->  tr[cc][r]
        }
        // a wave's LDS operations complete in order: no barrier needed for a wave-private buffer
        if (!hi) {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[32 + r] = tr[r * 33 + cc];  // G[32 + r][cc] = IJ[cc][r]
        }
    }
    float diag, bpiv;
    evdw_init_state(g, lane, diag, bpiv);
    float off0, offt;
    evdw_measure(g, diag, lane, I < kb, J < kb, off0, offt);
    const bool is_nan = off0 != off0;
    const bool rotate = !(is_nan || off0 < tol);
    if (lane == 0) {
        atomicMax(&maxoff_bits[b], is_nan ? 0x7fc00000u : __float_as_uint(offt));
        *act_flag = rotate ? 1 : 0;
        if (rotate && offt >= tol) atomicAdd(&nrot[b], 1);
        if (v3.hist && !is_nan) {  // debug: decade histogram of the pair measure (ASVD_DEBUG_HIST)
            const int bk = (int)floorf(-log10f(fmaxf(off0, 1e-30f)));
            atomicAdd(&v3.hist[bk < 0 ? 0 : (bk > 9 ? 9 : bk)], 1);
        }
    }
    float* d0 = KEEPG ? v3.Gd32 + ((int64_t)b * v3.nbpan + I) * 1024 : nullptr;
    float* d1 = KEEPG ? v3.Gd32 + ((int64_t)b * v3.nbpan + J) * 1024 : nullptr;
    if constexpr (NW > 1) {   // the helpers are waiting in coop_sweep whether this pair rotates or not
        evdw_identity(q, lane);
        coop_sweep<NW>(g, q, diag, bpiv, rotate, lane, 0, coopbuf);
    }
    if (!rotate) {
        if (KEEPG && v3.Gd32) {  // carried diagonal blocks of the two panels = the blocks of the matrix itself
            const int hi = lane >> 5, cc = lane & 31;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if (!hi) d0[r * 32 + cc] = g[r];
                else d1[r * 32 + cc] = g[32 + r];
            }
        }
        return;
    }
    if constexpr (NW == 1) {
        evdw_identity(q, lane);
        const int nsw = (off0 > 0.05f) ? inner_sweeps : min(1, inner_sweeps);
        evdw_sweep(g, q, diag, bpiv, lane, nsw * EVD_PHASE_PAIRS, csbuf[wv]);
    }
    float cs;
    int rnk;
    evdw_finish(q, diag, lane, true, cs, rnk);
    float* __restrict__ qo = Qbuf + slot * (PW * PW);
#pragma unroll
    for (int r = 0; r < 64; ++r) qo[r * PW + rnk] = q[r] * cs;
    if (KEEPG && v3.Gd32) evdw_store_diag_blocks(g, diag, cs, rnk, lane, d0, d1);
}

// --------------------------------------------------------------------------------------------------
// Both inner steps of a super-pair (S, T) in ONE launch: the work of evd_kernel<1,1> + evd_kernel<2,1>.  Two waves per super-pair; the four
// 32-blocks are S0, S1, T0, T1 = 0..3.
//   step 0: wave sp solves sub-pair (sp, 2 + sp): carried diagonal blocks of its two panels + the summed cross tile [0,2] / [1,3];
//           its sorted, rescaled eigenvectors Q0_sp (two 64 x 32 halves) and the two transformed diagonal blocks go to LDS;
//   step 1: wave sp solves sub-pair (sp, 3 - sp): diagonal blocks from step 0, cross block  Q0_sp[:, :32]^T MM Q0_(1-sp)[:, 32:]  with
//           MM = M (sp = 0) or M^T (sp = 1), M = G[{0,2},{1,3}] assembled from the other four tiles — two small fp32-MFMA products;
//           epilogue: the new carried blocks of its two panels (global) and its 128 x 64 column block of Qfin = Q^(0) Q^(1).
// The step-0 eigenvectors travel between the two waves through GLOBAL memory (v3.Q0, 16 KB per solve, L2 resident; agent-scope loads): with
// them in LDS the workgroup needed 50 KB = three workgroups per CU, and a launch of 1024 super-pairs ran as 768 + 256 workgroups with half
// of the SIMDs idle in the second round (measured 570 us per launch; two workgroups per CU: 700 us).  LDS now (33,536 B: four workgroups
// per CU, every SIMD holds two waves, ONE round):  one 64 x 65 region that is, in turn, the transposer of the step-0 images, the step-0
// diagonal blocks, the padded M and the staging of the cross blocks, and two wave-private [64][33] slices where the epilogue stages the
// Q0 half it multiplies.  Row strides 33 / 65 make every MFMA operand read (lanes along a row OR along a column) conflict free.
constexpr int QH_LD = 33, QH_FLOATS = 64 * QH_LD;   // one half of a Q0: 64 rows x 32 sorted columns
constexpr int M_LD = 65;
constexpr int E12_SLICE = 64 * M_LD;                // float offset of the two epilogue slices
constexpr int E12_CS = E12_SLICE + 2 * QH_FLOATS;   // float offset of the two row-coefficient buffers
constexpr int E12_SMEM_FLOATS = E12_CS + 2 * 128;

// agent-scope load of data another wave of the workgroup (or this wave) wrote to global memory earlier in the launch: L2, never this CU's L1
__device__ __forceinline__ float ldg_sc1(const float* p) { return __int_as_float(__hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

__device__ __forceinline__ int mfma_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }  // C/D row of v_mfma_f32_32x32x2_f32

// NW = 1: two waves per super-pair, each solve wave-local (throughput form).  NW = 4: eight waves, waves 0 / 1 are the MAIN waves of the two
// solves and run exactly the code of the NW = 1 form except that their sweeps are cooperative (coop_sweep) with the helper waves
// 2 h + sp, h = 1..3, which only take part in the sweeps and the workgroup barriers (latency form: one workgroup per CU).
// RING: both inner steps are cross-only visits on the ring (evdw_sweep_ring / coop_sweep<NW, true>): half the phases per solve.
template <int NW, int RINGM>   // RINGM: 0 full visits, 1 cross-only ring visits in both inner steps, 2 in step 1 only
__global__ __launch_bounds__(128 * NW, NW == 1 ? 2 : 1) void evdw12_kernel(Sched sc, unsigned* __restrict__ maxoff_bits, int* __restrict__ nrot,
                                                        const int* __restrict__ done, float tol, int inner_sweeps, int nb, int step, int kb,
                                                        EvdV3 v3, long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float e12_smem[];
    constexpr bool RING0 = RINGM == 1, RING1 = RINGM >= 1;
    // stage trace (ASVD_EVDW_TRACE=1): shader-clock stamps of workgroup (0, 0), one row of 16 per wave
    int tstage = 0;
    auto stamp = [&]() {
        if (trace && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) trace[(threadIdx.x >> 6) * 16 + tstage] = (long long)__builtin_amdgcn_s_memtime();
        ++tstage;
    };
    stamp();
    float* const R16 = e12_smem;                // multi-purpose region (see above)
    const int tid = threadIdx.x, lane = tid & 63, wv_ = __builtin_amdgcn_readfirstlane(tid >> 6), sp = wv_ & 1, hw = wv_ >> 1;
    const int hi = lane >> 5, cc = lane & 31;
    const int pair = blockIdx.x, b = blockIdx.y, npairs = gridDim.x;
    float* const LC = e12_smem + E12_SMEM_FLOATS + sp * Coop<NW == 1 ? 4 : NW>::FLOATS;   // cooperative-sweep block of this solve (NW > 1 only)
    if (ld_flag(done + b)) return;              // uniform over the workgroup
    const int64_t slot = (int64_t)b * npairs + pair;
    int* const sub = v3.subact + slot * 4;      // [step 0: sub-pairs 0, 1 | step 1: sub-pairs 0, 1]
    int S, T;
    super_pair(sc, v3.ns, step, pair, S, T);
    if (T >= v3.ns) {                           // padding super-pair
        if (lane == 0) { sub[sp] = 0; sub[2 + sp] = 0; }
        // a present lower member still streams through the fused update + Gram kernel (its panels feed the next step's tiles): its column
        // norms are the carried diagonals as they stand
        if (v3.Din && hw == 0) {
            float dv = 0.0f;
            if (S < v3.ns && lane < 32) dv = v3.Gd32[((int64_t)b * v3.nbpan + 2 * S + sp) * 1024 + lane * 33];
            v3.Din[slot * 128 + (lane < 32 ? 32 * sp + lane : 64 + 32 * sp + (lane - 32))] = dv;
        }
        return;
    }
    const float* __restrict__ gx = v3.Gx6 + slot * v3.nsplit6 * (6 * 1024);
    float g[64], q[64];
    float diag, bpiv, off0, offt, cs;
    int rnk;
    if constexpr (NW > 1) {
        if (hw > 0) {   // helper wave: the two cooperative sweeps and the barriers of the main waves in between, nothing else
            float dd = 0.0f;
            coop_sweep<NW, RING0>(g, q, dd, 0.0f, false, lane, hw, LC);
            __syncthreads(); __syncthreads();                       // end of step 0
            __syncthreads(); __syncthreads(); __syncthreads();      // step-1 image assembly
            coop_sweep<NW, RING1>(g, q, dd, 0.0f, false, lane, hw, LC);
            return;
        }
    }

    // ================================ step 0: sub-pair (sp, 2 + sp) ================================
    {
        const int I = 2 * S + sp, J = 2 * T + sp;
        const float* __restrict__ dA = v3.Gd32 + ((int64_t)b * v3.nbpan + I) * 1024;
        const float* __restrict__ dB = v3.Gd32 + ((int64_t)b * v3.nbpan + J) * 1024;
        const float* __restrict__ ct = gx + (sp ? 3 : 0) * 1024 + cc;   // cross tile [0,2] or [1,3]
        if (!hi) {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[r] = dA[r * 32 + cc];
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[r] = 0.0f;
#pragma unroll 2
            for (int s2 = 0; s2 < v3.nsplit6; ++s2) {   // partials in ascending order, 32 independent loads each
#pragma unroll
                for (int r = 0; r < 32; ++r) g[r] += ct[(int64_t)s2 * 6144 + r * 32];
            }
#pragma unroll
            for (int r = 0; r < 32; ++r) g[32 + r] = dB[r * 32 + cc];
            float* tr = R16 + sp * (32 * 33);            // wave-private transposer
#pragma unroll
            for (int r = 0; r < 32; ++r) tr[cc * 33 + r] = g[r];       // C[r][cc] -> tr[cc][r]
        }
        // lanes 32..63 wrote, lanes 0..31 read: one wave, so program order is the only order there is — pinned for the compiler (no motion of
        // the reads into the first `!hi` region above) and for the LDS queue
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!hi) {
            const float* tr = R16 + sp * (32 * 33);
#pragma unroll
            for (int r = 0; r < 32; ++r) g[32 + r] = tr[r * 33 + cc];  // G[32 + r][cc] = C[cc][r]
        }
        stamp();  // 1: image loaded
        evdw_init_state(g, lane, diag, bpiv);
        if (v3.Din) v3.Din[slot * 128 + (lane < 32 ? 32 * sp + lane : 64 + 32 * sp + (lane - 32))] = diag;   // pre-rotation squared column norms, Q order
        evdw_measure(g, diag, lane, I < kb, J < kb, off0, offt);
        const bool is_nan = off0 != off0;
        const bool rotate = !(is_nan || off0 < tol);
        if (lane == 0) {
            atomicMax(&maxoff_bits[b], is_nan ? 0x7fc00000u : __float_as_uint(offt));
            sub[sp] = rotate ? 1 : 0;
            if (rotate && offt >= tol) atomicAdd(&nrot[b], 1);
            if (v3.hist && !is_nan) {  // debug: decade histogram of the pair measure (ASVD_DEBUG_HIST)
                const int bk = (int)floorf(-log10f(fmaxf(off0, 1e-30f)));
                atomicAdd(&v3.hist[bk < 0 ? 0 : (bk > 9 ? 9 : bk)], 1);
            }
        }
        evdw_identity(q, lane);
        stamp();  // 2: measured
        asm volatile("" ::: "memory");
        if constexpr (NW > 1) coop_sweep<NW, RING0>(g, q, diag, bpiv, rotate, lane, 0, LC);
        else if (rotate) {
            const int nsw = (off0 > 0.05f) ? inner_sweeps : min(1, inner_sweeps);
            if constexpr (RING0) evdw_sweep_ring(g, q, diag, bpiv, lane, nsw, e12_smem + E12_CS + sp * 128);
            else evdw_sweep(g, q, diag, bpiv, lane, nsw * EVD_PHASE_PAIRS, e12_smem + E12_CS + sp * 128);
        }
        asm volatile("" ::: "memory");   // no load of a later stage is hoisted above the sweep (its registers would be spilled across it)
        stamp();  // 3: swept
        evdw_finish(q, diag, lane, rotate, cs, rnk);
        stamp();  // 4: norms + ranks
        // sorted, rescaled eigenvectors, row-major 64 x 64 in global memory: column `lane` goes to column rnk (a permutation inside a 256-byte row)
        float* __restrict__ qdst = v3.Q0 + (slot * 2 + sp) * (PW * PW) + rnk;
#pragma unroll
        for (int r = 0; r < 64; ++r) qdst[r * PW] = q[r] * cs;
        __syncthreads();   // both transposers are done with: the region becomes the diagonal blocks
        // transformed diagonal blocks: sorted positions 0..31 -> block sp, 32..63 -> block 2 + sp
        evdw_store_diag_blocks(g, diag, cs, rnk, lane, R16 + sp * 1024, R16 + (2 + sp) * 1024);
    }
    stamp();  // 5: Q0 + diagonal blocks stored
    __syncthreads();   // diagonal blocks of both solves are in LDS, both Q0 in global memory (stores acknowledged: workgroup-scope release)
    stamp();  // 6

    // ================================ step 1: sub-pair (sp, 3 - sp) ================================
    const int I1 = 2 * S + sp, J1 = 2 * T + (1 - sp);
    // the thread index is made opaque here: otherwise the address arithmetic of everything below is hoisted above the step-0 sweep and
    // its two dozen values are spilled across it (scratch must stay 0, DESIGN.md 3.8 a)
    int tid1 = threadIdx.x;
    asm volatile("" : "+v"(tid1));
    const int lane1 = tid1 & 63, hi1 = lane1 >> 5, cc1 = lane1 & 31;
    {
        const float* dA = R16 + sp * 1024;
        const float* dB = R16 + (3 - sp) * 1024;
        if (!hi1) {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[r] = dA[r * 32 + cc1];
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[32 + r] = dB[r * 32 + cc1];
        }
    }
    __syncthreads();   // diagonal blocks consumed: the region becomes M
    {
        // M = G[{0,2},{1,3}] = [[tile 4, tile 1], [tile 2 ^T, tile 5]] (tiles [0,1] [0,3] [1,2] [2,3]), padded row stride M_LD, summed over the
        // partials in ascending order; every quadrant is read in the linear order of its tile (coalesced)
        float mv[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) mv[e] = 0.0f;
        for (int s2 = 0; s2 < v3.nsplit6; ++s2) {   // partials outermost: 32 independent loads in flight per partial
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int tile = qd == 0 ? 4 : (qd == 1 ? 1 : (qd == 2 ? 2 : 5));
#pragma unroll
                for (int j = 0; j < 8; ++j) mv[qd * 8 + j] += gx[(int64_t)s2 * 6144 + tile * 1024 + tid1 + 128 * j];
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = tid1 + 128 * j, tr_ = e >> 5, tc = e & 31;
                const int mi = qd == 0 ? tr_ : (qd == 1 ? tr_ : (qd == 2 ? 32 + tc : 32 + tr_));
                const int mk = qd == 0 ? tc : (qd == 1 ? 32 + tc : (qd == 2 ? tr_ : 32 + tc));
                R16[mi * M_LD + mk] = mv[qd * 8 + j];
            }
        }
    }
    __syncthreads();
    stamp();  // 7: M assembled
    f32x16 cacc = {0};
    {
        // T = MM QBh (64 x 32, K = 64):  MM = M (sp 0) or M^T (sp 1);  QBh = second half of the OTHER solve's Q0
        const float* __restrict__ QB = v3.Q0 + (slot * 2 + (1 - sp)) * (PW * PW) + 32 + cc1;   // second half of the other solve's Q0
        const float* __restrict__ QA = v3.Q0 + (slot * 2 + sp) * (PW * PW) + cc1;              // first half of this solve's
        float bqv[32], qav[32];
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) bqv[k2] = ldg_sc1(QB + (2 * k2 + hi1) * PW);            // all operand loads in flight at once
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            qav[2 * reg] = ldg_sc1(QA + mfma_row(reg, hi1) * PW);
            qav[2 * reg + 1] = ldg_sc1(QA + (32 + mfma_row(reg, hi1)) * PW);
        }
        f32x16 t0 = {0}, t1 = {0};
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) {
            const int k = 2 * k2 + hi1;
            const float bq = bqv[k2];
            const float a0 = sp ? R16[k * M_LD + cc1] : R16[cc1 * M_LD + k];
            const float a1 = sp ? R16[k * M_LD + 32 + cc1] : R16[(32 + cc1) * M_LD + k];
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq, t1, 0, 0, 0);
        }
        // C' = QAh^T T (32 x 32, K = 64): the accumulators of T are the B operands as they are (the reduction index of an MFMA may be
        // permuted freely as long as both operands agree): register `reg` of tile rt holds rows k = 32 rt + mfma_row(reg, h)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qav[2 * reg], t0[reg], cacc, 0, 0, 0);
            cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qav[2 * reg + 1], t1[reg], cacc, 0, 0, 0);
        }
    }
    __syncthreads();   // M consumed: the region becomes the staging of the two cross blocks
    {
        float* Cs = R16 + sp * (32 * 33);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) Cs[mfma_row(reg, hi1) * 33 + cc1] = cacc[reg];   // C'[i][j], i = row, j = cc1
        // a wave's LDS operations complete in order
        if (hi1) {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[r] = Cs[r * 33 + cc1];          // G[r][32 + cc1] = C'[r][cc1]
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[32 + r] = Cs[cc1 * 33 + r];     // G[32 + r][cc1] = C'[cc1][r]
        }
    }
    stamp();  // 8: cross block computed, image complete
    evdw_init_state(g, lane1, diag, bpiv);
    evdw_measure(g, diag, lane1, I1 < kb, J1 < kb, off0, offt);
    stamp();  // 9: measured
    asm volatile("" ::: "memory");
    {
        const bool is_nan = off0 != off0;
        const bool rotate = !(is_nan || off0 < tol);
        if (lane1 == 0) {
            atomicMax(&maxoff_bits[b], is_nan ? 0x7fc00000u : __float_as_uint(offt));
            sub[2 + sp] = rotate ? 1 : 0;
            if (rotate && offt >= tol) atomicAdd(&nrot[b], 1);
            if (v3.hist && !is_nan) {  // debug: decade histogram of the pair measure (ASVD_DEBUG_HIST)
                const int bk = (int)floorf(-log10f(fmaxf(off0, 1e-30f)));
                atomicAdd(&v3.hist[bk < 0 ? 0 : (bk > 9 ? 9 : bk)], 1);
            }
        }
        evdw_identity(q, lane1);
        if constexpr (NW > 1) coop_sweep<NW, RING1>(g, q, diag, bpiv, rotate, lane1, 0, LC);
        else if (rotate) {
            const int nsw = (off0 > 0.05f) ? inner_sweeps : min(1, inner_sweeps);
            if constexpr (RING1) evdw_sweep_ring(g, q, diag, bpiv, lane1, nsw, e12_smem + E12_CS + sp * 128);
            else evdw_sweep(g, q, diag, bpiv, lane1, nsw * EVD_PHASE_PAIRS, e12_smem + E12_CS + sp * 128);
        }
        asm volatile("" ::: "memory");
        stamp();  // 10: swept
        evdw_finish(q, diag, lane1, rotate, cs, rnk);
    }
    stamp();  // 11
    // thread index opaque once more: the per-lane addresses of the epilogue are not computed (and kept live) before the step-1 sweep
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, hi2 = lane2 >> 5, cc2 = lane2 & 31;
    // new carried diagonal blocks of the two panels
    evdw_store_diag_blocks(g, diag, cs, rnk, lane2, v3.Gd32 + ((int64_t)b * v3.nbpan + I1) * 1024, v3.Gd32 + ((int64_t)b * v3.nbpan + J1) * 1024);

    stamp();  // (between 11 and 12): diagonal blocks stored
    // Qfin[:, columns of blocks (ba, bb)] = Q^(0)[:, {ba, bb}] Q1,  ba = sp, bb = 3 - sp:
    //   rows of blocks {ba, ba + 2}  <-  Q0_ba[:, :32]       Q1[:32, :]        (product 0)
    //   rows of blocks {bb - 2, bb}  <-  Q0_(bb-2)[:, 32:]   Q1[32:, :]        (product 1)
    // Q1 sits in registers, lane2 = (unsorted) column: one v_permlane32_swap per register pair (k, k + 1) turns it into the B operands of the
    // two column tiles (lanes 0..31: positions 0..31 | 32..63, half-waves = the two k of an MFMA).  Columns are put in sorted order by the
    // store address.
    {
        const int ba = sp, bb = 3 - sp;
#pragma unroll
        for (int r = 0; r < 64; ++r) q[r] *= cs;
        const int rnk_lo = __shfl(rnk, cc2, 64), rnk_hi = __shfl(rnk, 32 + cc2, 64);   // sorted index of positions cc2 and 32 + cc2
        const int col_lo = rnk_lo < 32 ? 32 * ba + rnk_lo : 32 * bb + (rnk_lo - 32);
        const int col_hi = rnk_hi < 32 ? 32 * ba + rnk_hi : 32 * bb + (rnk_hi - 32);
        float* __restrict__ qf = v3.Qfin + slot * (128 * 128);
#pragma unroll
        for (int prod = 0; prod < 2; ++prod) {
            // stage the 64 x 32 half this product multiplies: coalesced agent-scope loads, padded wave-private slice (a wave's LDS operations
            // complete in order: no barrier)
            float* A = e12_smem + E12_SLICE + sp * QH_FLOATS;                          // [64 rows][33], columns = the k of this product
            {
                const float* __restrict__ src = v3.Q0 + (slot * 2 + (prod ? bb - 2 : ba)) * (PW * PW) + (prod ? 32 : 0) + cc2;
                float hv[32];
#pragma unroll
                for (int it = 0; it < 32; ++it) hv[it] = ldg_sc1(src + (2 * it + hi2) * PW);
#pragma unroll
                for (int it = 0; it < 32; ++it) A[(2 * it + hi2) * QH_LD + cc2] = hv[it];
            }
            f32x16 d00 = {0}, d01 = {0}, d10 = {0}, d11 = {0};   // [row tile][column tile]
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int ra = 32 * prod + 2 * kk, rb = ra + 1;
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[ra]), __float_as_uint(q[rb]), false, false);
                const float b0 = __uint_as_float(sw[0]), b1 = __uint_as_float(sw[1]);   // column tiles 0 / 1, k = (ra | rb) by half-wave
                const int kl = 2 * kk + hi2;                                              // this half-wave's k within the product
                const float a0 = A[cc2 * QH_LD + kl], a1 = A[(32 + cc2) * QH_LD + kl];
                d00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, d01, 0, 0, 0);
                d10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, d10, 0, 0, 0);
                d11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, d11, 0, 0, 0);
            }
            const int rblk0 = prod ? bb - 2 : ba, rblk1 = prod ? bb : ba + 2;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = mfma_row(reg, hi2);
                qf[(32 * rblk0 + i) * 128 + col_lo] = d00[reg];
                qf[(32 * rblk0 + i) * 128 + col_hi] = d01[reg];
                qf[(32 * rblk1 + i) * 128 + col_lo] = d10[reg];
                qf[(32 * rblk1 + i) * 128 + col_hi] = d11[reg];
            }
        }
    }
    stamp();  // 12: Qfin written
}

// Test hook kernel: one wave per 64x64 symmetric matrix (row-major), `sweeps` full inner sweeps, outputs the UNSORTED eigenvector
// matrix Q [64][64] (row r, position c), the closed-form diagonal, the sort ranks, the column scales and the final image.
template <bool RING>
__global__ __launch_bounds__(64, 2) void evdw_test_kernel(const float* __restrict__ Gin, int sweeps, float* __restrict__ Qout,
                                                       float* __restrict__ diag_out, int* __restrict__ rnk_out, float* __restrict__ cs_out,
                                                       float* __restrict__ Gout, float* __restrict__ meas_out) {
    const int lane = threadIdx.x, b = blockIdx.x;
    __shared__ __attribute__((aligned(16))) float csbuf[EVDW_CS_FLOATS];
    float g[64], q[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) g[r] = Gin[(int64_t)b * 4096 + r * 64 + lane];
    float diag, bpiv;
    evdw_init_state(g, lane, diag, bpiv);
    float off0, offt;
    evdw_measure(g, diag, lane, true, false, off0, offt);
    if (lane == 0) { meas_out[2 * b] = off0; meas_out[2 * b + 1] = offt; }
    evdw_identity(q, lane);
    if constexpr (RING) evdw_sweep_ring(g, q, diag, bpiv, lane, sweeps, csbuf);      // cross-only visits on the ring (image returned in ring positions)
    else evdw_sweep(g, q, diag, bpiv, lane, sweeps * 32, csbuf);
    float cs;
    int rnk;
    evdw_finish(q, diag, lane, true, cs, rnk);
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        Qout[(int64_t)b * 4096 + r * 64 + lane] = q[r];
        Gout[(int64_t)b * 4096 + r * 64 + lane] = g[r];
    }
    diag_out[b * 64 + lane] = diag;
    rnk_out[b * 64 + lane] = rnk;
    cs_out[b * 64 + lane] = cs;
}

}  // namespace

namespace asvdk {

static int evdq_env() {
    static const int v = getenv("ASVD_EVDQ") ? atoi(getenv("ASVD_EVDQ")) : -1;   // 0 / 1 force the wave-local / cooperative launches
    return v;
}

void launch_evdw0(bool keepg, int npairs, int batch, hipStream_t st, const Sched& sc, const float* Gpart, int nsplit, float* Qbuf, int* active,
                  unsigned* maxoff_bits, int* nrot, const int* done, float tol, int inner_sweeps, int nb, int step, int kb, const int* plist,
                  int list_stride, const EvdV3& v3) {
    // latency form (one pair per workgroup, four waves per solve; bit-identical) when the pairs of the launch would leave most SIMDs idle
    const bool coop = inner_sweeps == 1 && (evdq_env() == 1 || (evdq_env() != 0 && (long long)npairs * batch <= 2 * call_cus_now()));
    if (coop) {
        const dim3 grid((unsigned)npairs, (unsigned)batch);
        if (keepg)
            evdw0_kernel<1, 4><<<grid, 256, 0, st>>>(sc, Gpart, nsplit, Qbuf, active, maxoff_bits, nrot, done, tol, inner_sweeps, nb, step, kb, plist,
                                                     list_stride, npairs, v3);
        else
            evdw0_kernel<0, 4><<<grid, 256, 0, st>>>(sc, Gpart, nsplit, Qbuf, active, maxoff_bits, nrot, done, tol, inner_sweeps, nb, step, kb, plist,
                                                     list_stride, npairs, v3);
        return;
    }
    const dim3 grid((unsigned)((npairs + 3) / 4), (unsigned)batch);
    if (keepg)
        evdw0_kernel<1, 1><<<grid, 256, 0, st>>>(sc, Gpart, nsplit, Qbuf, active, maxoff_bits, nrot, done, tol, inner_sweeps, nb, step, kb, plist, list_stride,
                                                 npairs, v3);
    else
        evdw0_kernel<0, 1><<<grid, 256, 0, st>>>(sc, Gpart, nsplit, Qbuf, active, maxoff_bits, nrot, done, tol, inner_sweeps, nb, step, kb, plist, list_stride,
                                                 npairs, v3);
}

int evdw12_lds_bytes() { return (int)(E12_SMEM_FLOATS * sizeof(float)); }
constexpr int EVDQ_NW = 4;
static int evdq12_lds_bytes() { return (int)((E12_SMEM_FLOATS + 2 * Coop<EVDQ_NW>::FLOATS) * sizeof(float)); }

void launch_evdw12(int npairs_s, int batch, hipStream_t st, const Sched& sc, unsigned* maxoff_bits, int* nrot, const int* done, float tol,
                   int inner_sweeps, int nb, int step, int kb, const EvdV3& v3, int sweep, int ring_default) {
    // the > 64 KB dynamic-LDS opt-in is per device: one flag per device the process drives (idempotent; a race only repeats the calls)
    static bool attr_done[64] = {};
    int devid = 0;
    (void)hipGetDevice(&devid);
    if (devid < 0 || devid >= 64 || !attr_done[devid]) {
        bool ok = true;
        ok = ok && hipFuncSetAttribute((const void*)evdw12_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, evdw12_lds_bytes()) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)evdw12_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, evdw12_lds_bytes()) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)evdw12_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, evdw12_lds_bytes()) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)evdw12_kernel<EVDQ_NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, evdq12_lds_bytes()) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)evdw12_kernel<EVDQ_NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, evdq12_lds_bytes()) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)evdw12_kernel<EVDQ_NW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, evdq12_lds_bytes()) == hipSuccess;
        if (ok && devid >= 0 && devid < 64) attr_done[devid] = true;
    }
    // latency form when the launch cannot even give every CU one workgroup: four waves per solve (bit-identical results).  It needs one
    // full inner sweep per visit (the default) and the standard 32 phase pairs.  ASVD_EVDQ=0 / 1 forces the choice.
    const bool coop = inner_sweeps == 1 && (evdq_env() == 1 || (evdq_env() != 0 && (long long)npairs_s * batch <= call_cus_now()));
    static long long* trace = nullptr;
    static int traced = 0;
    if (getenv("ASVD_EVDW_TRACE") && !trace) (void)hipMalloc(&trace, 16 * 16 * sizeof(long long));
    long long* tr = traced < 3 ? trace : nullptr;
    // cross-only visits on the ring (evdw_sweep_ring).  ring_default (the driver's choice): 2 = inner step 1 cross-only — its two panels arrive with
    // exactly diagonal Gram blocks, step 0 has just diagonalised them, so the visit loses nothing at its start — for problems of >= 2048 columns:
    // measured +2 % at 32 x 4096^2 (eigen-solves 108 -> 91.5 ms per step, a tiny ninth sweep for some problems), +5 % at 32 x 2048^2, -10 % on the
    // latency of a lone 4096^2 (95.5 -> 86.3 ms); 768-column problems lose 7 % (one more sweep of 12 super-panels) and keep full visits.  Cross-only
    // in BOTH steps (mode 1) does not converge in reasonable time: 10-11 sweeps, 8 of them dense (profiles/r5_ring.txt).
    // Measurement knobs: ASVD_RING=0 / 1 / 2 forces the mode, ASVD_RING_FROM=k applies it from dense sweep k on.
    const char* er = getenv("ASVD_RING");
    const char* ef = getenv("ASVD_RING_FROM");
    const int ring = er ? ((sweep >= (ef ? atoi(ef) : 0)) ? atoi(er) : 0) : ring_default;
    const dim3 grid((unsigned)npairs_s, (unsigned)batch);
#define ASVD_E12(NWV, RM, THREADS, LDS) evdw12_kernel<NWV, RM><<<grid, THREADS, LDS, st>>>(sc, maxoff_bits, nrot, done, tol, inner_sweeps, nb, step, kb, v3, tr)
    if (coop) {
        if (ring == 1) ASVD_E12(EVDQ_NW, 1, 128 * EVDQ_NW, evdq12_lds_bytes());
        else if (ring == 2) ASVD_E12(EVDQ_NW, 2, 128 * EVDQ_NW, evdq12_lds_bytes());
        else ASVD_E12(EVDQ_NW, 0, 128 * EVDQ_NW, evdq12_lds_bytes());
    } else {
        if (ring == 1) ASVD_E12(1, 1, 128, evdw12_lds_bytes());
        else if (ring == 2) ASVD_E12(1, 2, 128, evdw12_lds_bytes());
        else ASVD_E12(1, 0, 128, evdw12_lds_bytes());
    }
#undef ASVD_E12
    if (tr) {
        long long h[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 2; ++w) {   // the two main waves
            fprintf(stderr, "[evdw12 trace] launch %d (%s) wave %d stage deltas (clocks):", traced, coop ? "four waves per solve" : "wave-local", w);
            for (int i = 1; i < 14; ++i) fprintf(stderr, " %lld", h[w * 16 + i] - h[w * 16 + i - 1]);
            fprintf(stderr, "\n");
        }
        ++traced;
    }
}

void launch_evdw_test(int batch, hipStream_t st, const float* G, int sweeps, float* Q, float* diag, int* rnk, float* cs, float* Gout, float* meas) {
    // sweeps < 0: |sweeps| cross-only visits on the ring
    if (sweeps < 0) evdw_test_kernel<true><<<batch, 64, 0, st>>>(G, -sweeps, Q, diag, rnk, cs, Gout, meas);
    else evdw_test_kernel<false><<<batch, 64, 0, st>>>(G, sweeps, Q, diag, rnk, cs, Gout, meas);
}

}  // namespace asvdk
