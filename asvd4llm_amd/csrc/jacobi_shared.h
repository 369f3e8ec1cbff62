// jacobi_shared.h — definitions shared by the two translation units of the block-Jacobi SVD: svd_jacobi.hip (streaming kernels, host
// driver) and evd_wave.hip (the wave-local 64x64 eigen-solver, compiled with -fno-slp-vectorize: the SLP vectoriser packs its per-register
// FMAs into v_pk_fma_f32, which cannot take a DPP operand — 389 extra v_mov_b32_dpp + 224 register shuffles per phase pair).
#pragma once
#include "common.h"
#include <cstring>

namespace asvdk {

constexpr int PB = 32;       // panel width
constexpr int PW = 2 * PB;   // pair width
constexpr int EVD_PHASE_PAIRS = PW / 2;   // (A, B) phase pairs of one inner sweep of the 64x64 eigen-solve: one full odd-even cycle, every column pair meets once

// Pair ordering of one Jacobi sweep: nb (even) panels, nb-1 steps, pair k in [0, nb/2) of step `step`.
//  * default: XOR ordering over the panel count padded to a power of two P — step d = step+1 (d = 1..P-1) pairs every panel i
//    with i^d; pairs that touch a padding panel (J >= nb) are skipped.  Steps 1, 2, 3, ... meet the nearest neighbours first,
//    which on the norm-sorted, Cholesky-preconditioned matrices is where the coupling is: measured 10 -> 8 sweeps at 4096^2
//    and 14 -> 9 on the row-scaled wide layers against the round-robin tournament (CPU prototype at n = 1024: 8 -> 6).
//  * c_pair_order = 0 (ASVD_ORDER=rr, for A/B measurements): round-robin tournament (circle method), nb-1 steps of nb/2 pairs.
// Everything a kernel needs to know about the call's pair schedules travels BY VALUE in its argument list (under 300 bytes of kernarg): round 2
// kept these in __constant__ symbols rewritten by every call, so two concurrent calls with different shapes (a grouped 13B schedule next
// to an XOR one) overwrote each other's tables mid-flight.
//   pair_order  1 XOR (default), 0 round-robin (ASVD_ORDER=rr)
//   super_order 1 XOR, 2 grouped (below);  gm / gpair: group pairs per round / {gA, gB} of the grouped schedule
struct Sched {
    int pair_order, super_order, gm;
    int gb;         // grouped schedule: log2 of the group size G (1..4)
    int meas;       // measurement builds only (-DASVD_SG_TIMING, tools/bench_supgram.py): timing-only ablation bits of the fused kernel; 0 in the product
    signed char gpair[16][8][2];
};
static Sched default_sched() {
    Sched sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.pair_order = 1;
    sc.super_order = 1;
    return sc;
}

static __device__ __forceinline__ void rr_pair(const Sched& sc, int nb, int step, int k, int& I, int& J) {
    if (sc.pair_order) {  // pair space padded to the next power of two: callers skip pairs with J >= nb
        const int d = step + 1;
        const int h = 31 - __clz(d);  // highest set bit of d: i < i^d  <=>  bit h of i is clear
        I = ((k >> h) << (h + 1)) | (k & ((1 << h) - 1));
        J = I ^ d;
        return;
    }
    const int a = (k == 0) ? 0 : 1 + (k - 1 + step) % (nb - 1);
    const int pb = nb - 1 - k;
    const int b = 1 + (pb - 1 + step) % (nb - 1);
    I = a < b ? a : b;
    J = a < b ? b : a;
}

// Pair ordering at the SUPER-PANEL level of the two-level sweeps (twolevel.h).  `step` counts from 0; pairs with T >= ns (padding) are skipped
// by the callers.
//   super_order 1: XOR like the panel level, over the super-panel count padded to a power of two.  A padded schedule runs P-1 super-steps
//     with many empty slots (13B: 80 super-panels -> 127 steps, 37 % empty).
//   super_order 2: GROUPED schedule for counts that are not a power of two: groups of G = 2^gb super-panels (G = 2 .. 16, ns / G <= 16 groups):
//     XOR (d = 1..G-1) inside the groups, then the group pairs of a round-robin tournament over the groups, each for the G offsets s
//     (A_i <-> B_{i ^ s}): the nearest-neighbour-first order of the XOR schedule inside a group and inside a group pair, and G - 1 + rounds * G
//     super-steps.  With an EVEN number of groups every slot of every step is filled and the sweep has the minimum ns - 1 steps (80 super-panels:
//     G = 8, 10 groups, 79 steps; round 2-3 used G = 16 only: 5 groups, a bye per round, 95 steps; the padded XOR schedule: 127).  The host
//     picks the G with the fewest steps (svd_jacobi.hip, group_bits_for).  sc.gpair[round][m] = {gA, gB} (gA < gB), sc.gm = pairs per round.  (A plain round-robin tournament over
//     the super-panels — ns-1 full steps — was measured in round 2: sweeps 7-9 -> 8-12, the nearest-neighbour-first order is worth more.)
static __device__ __forceinline__ void super_pair(const Sched& sc, int ns, int step, int k, int& S, int& T) {
    if (sc.super_order == 1) {
        const int d = step + 1;
        const int h = 31 - __clz(d);
        S = ((k >> h) << (h + 1)) | (k & ((1 << h) - 1));
        T = S ^ d;
        return;
    }
    if (sc.super_order == 2) {
        const int gb = sc.gb, G = 1 << gb;
        if (step < G - 1) {
            if (k >= ns / 2) { S = ns; T = ns; return; }
            const int d = step + 1, h = 31 - __clz(d), g = k >> (gb - 1), kk = k & ((G >> 1) - 1);
            S = G * g + (((kk >> h) << (h + 1)) | (kk & ((1 << h) - 1)));
            T = S ^ d;
            return;
        }
        const int r = (step - (G - 1)) >> gb, sft = (step - (G - 1)) & (G - 1), m = k >> gb, i = k & (G - 1);
        if (m >= sc.gm) { S = ns; T = ns; return; }
        S = G * sc.gpair[r][m][0] + i;
        T = G * sc.gpair[r][m][1] + (i ^ sft);
        return;
    }
    S = ns; T = ns;   // no other order exists
}

// Pair handled by a workgroup: from the schedule (plist == nullptr) or, in sparse sweeps, from an explicit per-problem list of
// marked pairs (code = I << 16 | J, -1 = empty slot).  Returns false when there is nothing to do for this slot.
static __device__ __forceinline__ bool get_pair(const Sched& sc, const int* __restrict__ plist, int list_stride, int b, int nb, int step, int pair, int& I, int& J) {
    if (plist) {
        const int code = ld_flag(plist + b * list_stride + pair);
        if (code < 0) return false;
        I = code >> 16;
        J = code & 0xffff;
        return true;
    }
    rr_pair(sc, nb, step, pair, I, J);
    return J < nb;  // padding pair of the XOR ordering
}

// --------------------------------------------------------------------------------------------------
// rotation of a 2x2 pivot block (a b; b d): c, s, t = tan.  Shared by the LDS solver (evd_body) and the wave-local one (evd_wave.h).
static __device__ __forceinline__ void jacobi_rot(float a, float d, float b, float& c, float& s, float& t) {
    // branch-free: b == 0 gives zeta = +-inf -> t = 0, c = 1, s = 0 by itself; a or d <= 0 (empty column) is masked at the end
    const float cosv = b * __builtin_amdgcn_rsqf(a) * __builtin_amdgcn_rsqf(d);  // |cos| of the two columns
    const float zeta = (d - a) * __builtin_amdgcn_rcpf(2.0f * b);
    float tt = copysignf(1.0f, zeta) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f)));
    float cc = __builtin_amdgcn_rsqf(fmaf(tt, tt, 1.0f));
    float ss = tt * cc;
    // unit-norm correction: delta = c^2 + s^2 - 1 via FMAs is accurate far below one ulp, so after scaling by (1 - delta/2)
    // only the unbiased rounding of c and s themselves remains (no systematic norm drift; the hardware rcp/rsq
    // approximations above only perturb the ANGLE, which the next visit corrects).
    const float hd = 0.5f * fmaf(ss, ss, fmaf(cc, cc, -1.0f));
    cc = fmaf(-cc, hd, cc);
    ss = fmaf(-ss, hd, ss);
    const bool rot = fabsf(cosv) > 1e-8f;  // false for NaN (zero / negative diagonal) as well
    c = rot ? cc : 1.0f;
    s = rot ? ss : 0.0f;
    t = rot ? tt : 0.0f;
}

// Buffers of the two-level sweeps as the eigen-solve kernels see them (see evd_body in svd_jacobi.hip for the modes).
struct EvdV3 {
    int ns;               // super-panels per problem
    int nbpan;            // 32-column panels per problem (stride of Gd32)
    const float* Gx6;     // sgram6 partials [slot][nsplit][6][32*32]
    int nsplit6;
    float* Gd32;          // carried diagonal blocks [problem][panel][32*32]
    float* Q0;            // [slot][2][64*64]
    float* D0;            // [slot][4][32*32]
    float* Qfin;          // [slot][128*128]
    int* subact;          // [slot][4]: step 0 sub-pairs 0,1; step 1 sub-pairs 0,1
    int* hist;            // debug (ASVD_DEBUG_HIST): 10 counters, decade histogram of the pair measures of a sweep; nullptr otherwise
    float* Din;           // [slot][128]: squared column norms of the super-pair BEFORE its rotation (carried diagonal), in Q order (blocks S0 S1 T0 T1):
                          //              the power-of-two column scales of the split-fp16 update (twolevel.h, supgram_kernel) come from these
};

// CUs the current call may use (0 = the whole device, 256): set by asvd_svd_batched per host thread when it runs a batch as two halves on
// CU-masked streams; the launch-form switches of the eigen-solver follow it
extern thread_local int g_call_cus;
int device_cus();     // CUs of the current device (256 when none is visible)
int call_cus_now();   // g_call_cus, or device_cus() when the call owns the whole device

// ---- launchers of the wave-local eigen-solver (evd_wave.hip) -------------------------------------------------------------------
// single-level solves: the work of evd_kernel<0, KEEPG> (grid: npairs x batch), one wave per pair
void launch_evdw0(bool keepg, int npairs, int batch, hipStream_t st, const Sched& sc, const float* Gpart, int nsplit, float* Qbuf, int* active,
                  unsigned* maxoff_bits, int* nrot, const int* done, float tol, int inner_sweeps, int nb, int step, int kb, const int* plist,
                  int list_stride, const EvdV3& v3);
// both inner steps of every super-pair of super-step `step` (the work of evd_kernel<1,1> + evd_kernel<2,1>), two waves per super-pair
void launch_evdw12(int npairs_s, int batch, hipStream_t st, const Sched& sc, unsigned* maxoff_bits, int* nrot, const int* done, float tol,
                   int inner_sweeps, int nb, int step, int kb, const EvdV3& v3, int sweep = 0, int ring_default = 0);
int evdw12_lds_bytes();
// test hook
void launch_evdw_test(int batch, hipStream_t st, const float* G, int sweeps, float* Q, float* diag, int* rnk, float* cs, float* Gout, float* meas);

}  // namespace asvdk
