// svd_jacobi.hip — K4: batched economy SVD in fp32 by one-sided block Jacobi, written for gfx950.  HOST DRIVER (plans, launch sequences,
// convergence logic, C entry points); the device kernels are in jacobi_kernels.h / twolevel.h / tall_kernels.h (this translation unit) and
// evd_wave.hip (the 64x64 eigen-solves, their own translation unit).
//
// Replaces the factorisation at modules/svd_linear.py:65 (oracle: torch.linalg.svd, BASELINE.json) and the values-only torch.svd at
// sensitivity.py:101.  See DESIGN.md 3 for the algorithm and the roofline accounting.
//
// Data layout in HBM (per problem): the oriented matrix (rows >= cols) is stored as nb = ncols_pad/32 column panels.  Panel I is a dense
// [R][32] fp32 array (row stride 128 B = one cache line), so a wave reads two consecutive rows of a panel with ONE fully coalesced 256-B
// load, and a panel pair is two contiguous streams.  Rows [0, m_pad) hold A (times the column scale); rows [m_pad, m_pad + n_pad) exist
// only when the right vectors of a DIRECT call are wanted: the backsolve at the end writes them there (they are never rotated).
//
// A call:  [reduce: fp64 Gram + sorted Cholesky-QR, Jacobi then runs on R^T — svd_tall]  ->  sweeps (svd_direct_run)  ->  finalize.
// A dense sweep (two-level, twolevel.h): the internal step d = 1 with the single-level kernels, then per super-step TWO launches — evdw12
// (all 64x64 solves of the step) and supgram (update + next step's Gram tiles).  A sparse sweep: coupling snapshot, marked pairs packed into
// rounds of disjoint pairs, three single-level launches per round (gram_kernel, evdw0, update_kernel).
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
#include <thread>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <fcntl.h>
#include <unistd.h>

namespace asvdk {
thread_local int g_call_cus = 0;   // see call_cus(): set per host thread for the duration of a half-batch call (0: the whole device)
// CUs of the current device (hipDeviceAttributeMultiprocessorCount, cached per device).  256 — the MI355X — when no device is visible: the
// host-side size queries (asvd_svd_worksize) answer on a CPU-only box too.
int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return 256;
    cache[dev].store(ncu, std::memory_order_relaxed);
    return ncu;
}
int call_cus_now() { return g_call_cus > 0 ? g_call_cus : device_cus(); }
}
namespace {
using namespace asvdk;


#include "jacobi_kernels.h"
#include "tall_kernels.h"
#include "gram_i8.h"
#include "nn_gemm_i8.h"
#include "snapshot_i8.h"

// --------------------------------------------------------------------------------------------------
// ASVD_ORDER=rr: round-robin tournament instead of the XOR pair schedule (A/B measurements; single-level sweeps only)
static bool pair_order_xor() {
    const char* e = getenv("ASVD_ORDER");
    return !(e && !strncmp(e, "rr", 2));
}

struct Plan {
    int batch;
    int64_t m, n;         // as given
    int transposed;       // oriented = A^T when m < n
    int rows, cols;       // oriented dims, rows >= cols
    int m_pad, n_pad, nb, npairs, R, R_upd, want_v, vmode;  // vmode: 0 no right vectors, 2 backsolve at the end (1, accumulate V in the sweeps, is gone)
    int nsplit, rows_per_split, rows_per_wg, nchunks;
    // two-level dense sweeps (twolevel.h): ns super-panels of 64 columns, npairs_s pair slots per super-step (power-of-two padded)
    int two, ns, npairs_s, nsplit_s, rows_per_split_s, nchunks_s, rows_per_wg_s, nchunks_q, rows_per_wg_q;
    size_t off_gd32, off_gx6, off_q0, off_d0, off_qfin, off_subact, off_din;
    int64_t panel_stride, batch_stride;
    // workspace offsets in bytes
    size_t off_x, off_xorig, off_gpart, off_q, off_active, off_sig, off_ina, off_inv, off_perm, off_flags, off_pflag, off_plist, off_snap, off_snapex, total;
};

// CUs the launches of the current call may use: those of the device (256 on the MI355X), or those of one half when asvd_svd_batched runs a batch
// as two halves on CU-masked streams (below).  The launch geometry of a plan (row splits, chunk counts) is sized for this many.
static int64_t call_cus() { return call_cus_now(); }

int make_plan(int batch, int64_t m, int64_t n, int want_u, int want_vv, Plan& p) {
    if (batch < 1 || m < 1 || n < 1 || m > (1 << 24) || n > (1 << 24)) return ASVD_E_BADARG;
    const int64_t cu = call_cus();
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
    p.vmode = want_vv ? 2 : 0;
    p.R = p.m_pad + (want_vv ? p.n_pad : 0);   // the V rows of a panel are written by the backsolve at the end, never rotated
    p.R_upd = p.m_pad;
    // gram: choose the row split that minimises (rounds of resident workgroups) x (chunks per wave + fixed overhead).
    // 3 workgroups of 4 waves fit per CU (148 VGPR+AGPR) -> 768 slots; a grid of 1.3 x slots costs 2 full rounds.
    {
        const int launch_batch = batch;  // problems per launch
        const int64_t nchunk_total = p.m_pad / 32;
        int64_t best_ns = 1;
        double best_cost = 1e300;
        for (int64_t ns = 1; ns <= nchunk_total && ns <= 64; ++ns) {
            const int64_t wgs = ns * p.npairs * launch_batch;
            const int64_t rounds = ceil_div64(wgs, 3 * cu);
            const int64_t chunks_wg = ceil_div64(nchunk_total, ns);
            const int64_t chunks_wave = ceil_div64(chunks_wg, 4);
            const double cost = (double)rounds * ((double)chunks_wave + 1.5) + 0.02 * ns;  // mild penalty: partials traffic
            if (cost < best_cost) { best_cost = cost; best_ns = ns; }
        }
        p.rows_per_split = (int)(ceil_div64(nchunk_total, best_ns) * 32);
        p.nsplit = (int)ceil_div64(p.m_pad, p.rows_per_split);
    }
    // update: 128-row iterations; aim for >= 1024 workgroups but >= 2 iterations per workgroup when possible
    int64_t wantc = ceil_div64(4 * cu, (int64_t)p.npairs * batch);
    int64_t iters_total = ceil_div64(p.R_upd, 128);
    int64_t nc = wantc < 1 ? 1 : (wantc > iters_total ? iters_total : wantc);
    p.rows_per_wg = (int)(ceil_div64(iters_total, nc) * 128);
    p.nchunks = (int)ceil_div64(p.R_upd, p.rows_per_wg);
    {
        // two-level dense sweeps: default for >= 8 panels under the XOR ordering (ASVD_TWOLEVEL=0 restores the single-level sweep)
        const char* e2 = getenv("ASVD_TWOLEVEL");
        p.two = (pair_order_xor() && p.nb >= 8 && !(e2 && atoi(e2) == 0)) ? 1 : 0;
        p.ns = p.nb / 2;
        int pw2 = 2;
        while (pw2 < p.ns) pw2 <<= 1;
        p.npairs_s = pw2 / 2;
        const int launch_batch = batch;
        // sgram6: 3 workgroups (32 KiB LDS, ~150 VGPRs) per CU -> 768 slots; 16-row chunks per wave, same cost model as the single-level Gram
        const int64_t nchunk_total = p.m_pad / 32;
        int64_t best_ns = 1;
        double best_cost = 1e300;
        for (int64_t ns = 1; ns <= nchunk_total && ns <= 64; ++ns) {
            const int64_t wgs = ns * p.npairs_s * launch_batch;
            const int64_t rounds = ceil_div64(wgs, 3 * cu);
            const int64_t chunks_wave = ceil_div64(2 * ceil_div64(nchunk_total, ns), 4);
            const double cost = (double)rounds * ((double)chunks_wave + 1.5) + 0.02 * ns;
            if (cost < best_cost) { best_cost = cost; best_ns = ns; }
        }
        p.rows_per_split_s = (int)(ceil_div64(nchunk_total, best_ns) * 32);
        p.nsplit_s = (int)ceil_div64(p.m_pad, p.rows_per_split_s);
        // supdate: 32-row tiles; aim for >= 1024 workgroups per launch and >= 4 tiles per workgroup
        const int64_t tiles = ceil_div64(p.R_upd, 32);
        int64_t wantc = ceil_div64(4 * cu, (int64_t)p.npairs_s * launch_batch);
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
        int64_t nq = std::max<int64_t>(1, std::min<int64_t>(ceil_div64(cu, quads), std::max<int64_t>(1, tiles / 8)));
        {
            // counts that are not a power of two (13B: 80 super-panels = 20 real quads per problem in a 32-quad grid): one workgroup per CU means
            // the launch runs in whole rounds of 256 workgroups, and 320 real workgroups cost two rounds (measured 1179 us per launch at
            // 16 x 5120^2; four row chunks = exactly five rounds of a quarter length: 737 us).  Pick the chunk count that wastes the least.
            const int64_t real = ceil_div64(p.ns, 4) * batch;
            if (real != quads && real >= cu) {
                double best = 1e300;
                for (int64_t c = 1; c <= 8 && c <= std::max<int64_t>(1, tiles / 8); c *= 2) {
                    const double cost = (double)ceil_div64(real * c, cu) / (double)c * (1.0 + 0.03 * (double)c);
                    if (cost < best) { best = cost; nq = c; }
                }
            }
        }
        p.rows_per_wg_q = (int)(ceil_div64(tiles, nq) * 32);
        p.nchunks_q = (int)ceil_div64(p.R_upd, p.rows_per_wg_q);
    }
    p.panel_stride = (int64_t)p.R * PB;
    p.batch_stride = p.panel_stride * p.nb;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    p.off_x = take((size_t)p.batch_stride * batch * sizeof(float));
    p.off_xorig = take(p.vmode == 2 ? (size_t)p.m_pad * PB * p.nb * batch * sizeof(float) : 0);
    p.off_gpart = take((size_t)batch * p.npairs * p.nsplit * 3072 * sizeof(float));
    p.off_q = take((size_t)batch * p.npairs * PW * PW * sizeof(float));
    p.off_active = take((size_t)batch * p.npairs * sizeof(int));
    p.off_sig = take((size_t)batch * p.n_pad * sizeof(float));
    p.off_ina = take((size_t)batch * p.n_pad * sizeof(float));
    p.off_inv = take((size_t)batch * p.n_pad * sizeof(float));
    p.off_perm = take((size_t)batch * p.n_pad * sizeof(int));
    p.off_flags = take((size_t)batch * 4 * sizeof(int));  // [maxoff bits | nrot | done | super-pair updates] x batch (SoA)
    p.off_pflag = take((size_t)batch * p.nb * p.nb);      // sparse-sweep pair marks
    p.off_plist = take((size_t)batch * p.nb * p.nb * sizeof(int));  // per-step lists of marked pairs (bound: steps x nb/2 slots per problem)
    // int8 digit planes + column exponents of the coupling snapshot (snapshot_i8.h): problems that can reach a sparse sweep and fit the int32 range
    const bool snap8 = p.nb >= 8 && p.m_pad <= 32768;
    p.off_snap = take(snap8 ? (size_t)3 * p.n_pad * round_up64(p.m_pad, 64) * batch : 0);
    p.off_snapex = take(snap8 ? (size_t)batch * p.n_pad * sizeof(int) : 0);
    const size_t t2 = p.two ? 1 : 0;
    p.off_gd32 = take(t2 * batch * p.nb * 1024 * sizeof(float));                          // carried 32x32 diagonal blocks, one per panel
    p.off_gx6 = take(t2 * batch * p.npairs_s * std::max(p.nsplit_s, p.nchunks_q) * 6 * 1024 * sizeof(float));  // sgram6 / supgram partial tiles
    p.off_q0 = take(t2 * batch * p.npairs_s * 2 * PW * PW * sizeof(float));               // Q of the two step-0 solves
    p.off_d0 = take(t2 * batch * p.npairs_s * 4 * 1024 * sizeof(float));                  // diagonal blocks after step 0
    p.off_qfin = take(t2 * batch * p.npairs_s * SP * SP * sizeof(float));                 // Q^(0) Q^(1) of every super-pair
    p.off_subact = take(t2 * batch * p.npairs_s * 4 * sizeof(int));
    p.off_din = take(t2 * batch * p.npairs_s * SP * sizeof(float));                        // pre-rotation squared column norms of every super-pair
    p.total = off;
    return ASVD_OK;
}

// grouped schedule (super_pair, super_order = 2): the group size G = 2^gb, 2 <= G <= 16, that divides ns into at most 16 groups with the fewest
// super-steps, G - 1 + G * rounds (rounds of a round-robin tournament over the groups: ng - 1 for an even group count, ng with a bye for an odd one);
// ties go to the larger group.  0: ns is a power of two, or nothing beats the padded XOR schedule.
static int group_bits_for(int ns) {
    if (ns < 4 || (ns & (ns - 1)) == 0) return 0;
    int pw2 = 2;
    while (pw2 < ns) pw2 <<= 1;
    int best = pw2 - 1, best_gb = 0;
    // groups of 16 first, as rounds 2-3 had them: with the 13B shapes' 80 super-panels the 95 steps of five groups of 16 (one with a bye per
    // round: 32 pairs per step = exactly one round of eigen-solve waves at batch 32) beat the 79 steps of ten groups of 8 (40 pairs per step,
    // one more dense sweep: 22.4 vs 27.1 SVD/s at 5120^2 x 32) — locality inside a group is worth more than a full schedule
    if (ns % 16 == 0 && ns / 16 <= 8 && 15 + 16 * (((ns / 16) & 1) ? ns / 16 : ns / 16 - 1) < pw2 - 1) return 4;
    for (int gb = 4; gb >= 1; --gb) {
        const int G = 1 << gb;
        if (ns % G) continue;
        const int ng = ns / G;
        if (ng < 2 || ng > 16) continue;
        const int steps = G - 1 + G * ((ng & 1) ? ng : ng - 1);
        if (steps < best) { best = steps; best_gb = gb; }
    }
    return best_gb;
}
static int grouped_rounds(int ns, int gb) { const int ng = ns >> gb; return (ng & 1) ? ng : ng - 1; }
static int grouped_steps(int ns, int gb) { return (1 << gb) - 1 + (1 << gb) * grouped_rounds(ns, gb); }
// group pairs of every round: circle method over the ns / G groups (+ a bye when their number is odd)
static void set_group_table(Sched& sc, int ns, int gb) {
    const int ng = ns >> gb, n = ng + (ng & 1), gm = ng / 2;
    signed char (*tab)[8][2] = sc.gpair;
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
    sc.gb = gb;
}
static bool grouped_applies(int ns) { return group_bits_for(ns) > 0; }
static bool super_grouped_for(const Plan& p) {
    return p.two && grouped_applies(p.ns);
}

// ---- optional per-class timing with HIP events on the call's stream ------------------------------
// profiling state is per host thread: concurrent calls from different threads (on their own streams and workspaces) do not share it
thread_local bool g_prof_enabled = false;
thread_local int g_prof_mode = 0;   // asvd_svd_set_profiling: 0 off, 1 profile (the call runs UNSPLIT: every kernel alone on the chip), 2 profile and keep the split
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
// split-mode profile (mode 2): both halves time their own launches with events on their own stream; `g_prof_base` is an event recorded on the
// caller's stream before the halves start, so that the [start, end] of every fused update + Gram launch (class 8) of both halves can be placed
// on ONE time axis — the union / overlap of the two halves' HBM-bound launches is what the combined streaming rate of a split step follows from
thread_local hipEvent_t g_prof_base = nullptr;
thread_local std::vector<std::pair<float, float>> g_prof_iv8;
thread_local bool g_prof_last_split = false;
thread_local float g_prof_half_ms[2][9] = {{0}};
thread_local int g_prof_half_launches[2][9] = {{0}};
thread_local float g_prof_overlap[4] = {0, 0, 0, 0};   // class 8: summed ms of half 0, of half 1, union of both, time BOTH were in it
// which path the last asvd_svd_batched call of this host thread took (asvd_svd_get_last_path): ASVD_PATH_* bits of include/asvd_hip.h
thread_local int g_last_path = 0;

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
    g_prof_iv8.clear();
}
void prof_end() {
    for (auto& r : g_prof_recs) {
        float ms = 0;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        g_prof_ms[r.cls] += ms;
        g_prof_launches[r.cls] += 1;
        if (r.cls == 8 && g_prof_base) {
            float t0 = 0;
            if (hipEventElapsedTime(&t0, g_prof_base, r.a) == hipSuccess) g_prof_iv8.emplace_back(t0, t0 + ms);
        }
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

void asvd_svd_set_profiling(int enabled) { g_prof_enabled = enabled != 0; g_prof_mode = enabled < 0 ? 0 : (enabled > 2 ? 1 : enabled); }
int asvd_svd_get_split_profile(float* ms_host, int* launches_host, float* overlap_host) {
    if (!ms_host || !launches_host || !overlap_host) return ASVD_E_BADARG;
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < NPROF; ++i) { ms_host[h * NPROF + i] = g_prof_half_ms[h][i]; launches_host[h * NPROF + i] = g_prof_half_launches[h][i]; }
    for (int i = 0; i < 4; ++i) overlap_host[i] = g_prof_overlap[i];
    return g_prof_last_split ? 1 : 0;
}
// CUs the calls of THIS host thread may use (0 = the whole device): a caller that runs asvd_svd_batched on a CU-masked stream of its own says so
// here, so that the launch geometry (row splits, chunk counts, launch forms) is sized for those CUs; such calls are never split again.
void asvd_svd_set_call_cus(int cus) { g_call_cus = cus > 0 ? cus : 0; }
int asvd_svd_get_sweep_times(float* ms_host, long long* rotated_host, int cap) {
    const int n = (int)g_prof_sweep_ms.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (ms_host) ms_host[i] = g_prof_sweep_ms[i];
        if (rotated_host) rotated_host[i] = g_prof_sweep_rot[i];
    }
    return n;
}

int asvd_svd_get_last_path(void) { return g_last_path; }

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

// workspace of ONE call of `batch` problems whose launches are sized for the CUs of the current thread (call_cus())
static int worksize_one(int batch, int64_t m, int64_t n, int want_vectors, size_t* bytes) {
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

// ---- a batch as two halves on disjoint halves of the chip --------------------------------------------------------------------------------
// The two launches that alternate through a dense sweep use complementary resources — the eigen-solves the VALUs (no HBM), the fused update +
// Gram kernel the HBM path (VALUs nearly idle) — but they cannot share a CU (2 x 216 + 215 VGPRs per SIMD against the 512 of the register
// file), and two half-batches that share the WHOLE chip fall into lock-step (each kernel fills every CU as soon as it is free): 608 vs 628 ms.
// What does overlap them is SPACE: every half gets its own 128 CUs (hipExtStreamCreateWithCUMask) and its own host thread, the halves
// drift apart, and while one is in a compute-bound phase (eigen-solves, fp64 Cholesky-QR, snapshots, the long-side GEMM) the other's
// HBM-bound launches have the memory system to themselves — the fused kernel reaches 72 % of its full-chip throughput on half of the CUs
// (it saturates the HBM path at ~190 CUs: profiles/r5_cu_mask_scaling.jsonl).  Measured, 32 x 4096^2: 628 -> 595 ms (-5.3 %); 11008 x 4096
// and 4096 x 11008 x 32: -5 %; 2048-column problems: nothing (profiles/r5_ab_two_streams.jsonl).  So: batches of >= 4 problems with
// >= 3072 columns are split unless ASVD_SPLIT=0; a profiled call (asvd_svd_set_profiling) runs unsplit, so that the per-class
// durations describe each kernel alone on the chip.  The halves are ordinary calls with disjoint workspaces and outputs (the
// concurrency contract of the library), sized for 128 CUs; results are those of two half-batch calls.
// What the split owns (include/asvd_hip.h, "Ownership"): per CALLING HOST THREAD and device, created at that thread's first split call and
// released when the thread ends — two CU-masked streams and ONE worker thread (the second half runs on it; the first half on the calling
// thread).  Per thread, not per process: two host threads that make split calls at the same time each have their own pair of streams, so their
// calls stay independent and — what tests/test_gpu_concurrency.py holds the library to — whether a call is split never depends on timing.
struct SplitCtx {
    hipStream_t s[2] = {nullptr, nullptr};
    int dev = -1;
    int cus = 0;            // CUs of one half
    bool ok = false;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, job_done = false, stop = false;
    std::thread worker;
    ~SplitCtx() {
        if (worker.joinable()) {
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            cv.notify_all();
            worker.join();
        }
        for (int h = 0; h < 2; ++h)
            if (s[h]) (void)hipStreamDestroy(s[h]);
    }
};
struct SplitTls { std::vector<std::unique_ptr<SplitCtx>> v; };
static thread_local SplitTls g_split_tls;
static std::mutex g_split_mutex;
static std::atomic<int> g_split_mode{-1};   // asvd_svd_set_split: -1 environment / automatic, 0 never, 1 automatic

static void split_worker(SplitCtx* c) {
    (void)hipSetDevice(c->dev);
    std::unique_lock<std::mutex> lk(c->m);
    for (;;) {
        c->cv.wait(lk, [&] { return c->has_job || c->stop; });
        if (c->stop) return;
        std::function<void()> job = std::move(c->job);
        c->has_job = false;
        lk.unlock();
        try { job(); } catch (...) {}   // the job reports through the variables it captured; nothing may cross the thread boundary
        lk.lock();
        c->job_done = true;
        c->cv.notify_all();
    }
}

static SplitCtx* split_ctx() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    for (auto& c : g_split_tls.v)
        if (c->dev == dev) return c->ok ? c.get() : nullptr;
    g_split_tls.v.emplace_back(new SplitCtx());
    SplitCtx* c = g_split_tls.v.back().get();
    c->dev = dev;
    const int ncu = device_cus();
    if (ncu < 64 || (ncu & 1)) return nullptr;
    const int words = (ncu + 31) / 32;
    bool ok = true;
    for (int h = 0; h < 2 && ok; ++h) {
        std::vector<uint32_t> mask((size_t)words, 0u);
        for (int cu = h * (ncu / 2); cu < (h + 1) * (ncu / 2); ++cu) mask[cu >> 5] |= 1u << (cu & 31);
        ok = hipExtStreamCreateWithCUMask(&c->s[h], (uint32_t)words, mask.data()) == hipSuccess;
    }
    c->cus = ncu / 2;
    if (ok) {
        try { c->worker = std::thread(split_worker, c); } catch (...) { ok = false; }
    }
    c->ok = ok;
    return ok ? c : nullptr;
}

// Presence of this process on a device: one byte of /dev/shm/asvd_hip_split.<pci bus id>, taken as a POSIX record lock at the first
// asvd_svd_batched call on that device (the lock is gone when the process exits, however it exits; the descriptor stays open for the life of the
// process).  F_GETLK reports only locks of OTHER processes: that is how two ranks on one GPU (`--same_gpu`) see each other — both would otherwise
// claim "the first half + the second half" of the CUs and oversubscribe every one of them.
static int g_presence_fd[64];
static std::atomic<int> g_presence_state[64];   // 0 not tried, 1 registered, 2 unavailable (no /dev/shm, no bus id): treated as alone
static void presence_register(int dev) {
    if (dev < 0 || dev >= 64 || g_presence_state[dev].load(std::memory_order_acquire) != 0) return;
    std::lock_guard<std::mutex> lk(g_split_mutex);
    if (g_presence_state[dev].load(std::memory_order_relaxed) != 0) return;
    int state = 2;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, dev) == hipSuccess) {
        for (char* q = bus; *q; ++q) if (*q == ':' || *q == '/') *q = '_';
        char path[128];
        snprintf(path, sizeof(path), "/dev/shm/asvd_hip_split.%s", bus);
        const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd >= 0) {
            bool mine = false;
            for (int slot = 0; slot < 256 && !mine; ++slot) {
                struct flock fl {};
                fl.l_type = F_WRLCK; fl.l_whence = SEEK_SET; fl.l_start = slot; fl.l_len = 1;
                mine = fcntl(fd, F_SETLK, &fl) == 0;
            }
            if (mine) { g_presence_fd[dev] = fd; state = 1; } else close(fd);
        }
    } else (void)hipGetLastError();
    g_presence_state[dev].store(state, std::memory_order_release);
}
static bool split_device_shared(int dev) {
    if (dev < 0 || dev >= 64 || g_presence_state[dev].load(std::memory_order_acquire) != 1) return false;
    struct flock fl {};
    fl.l_type = F_WRLCK; fl.l_whence = SEEK_SET; fl.l_start = 0; fl.l_len = 256;
    if (fcntl(g_presence_fd[dev], F_GETLK, &fl) != 0) return false;
    return fl.l_type != F_UNLCK;
}

// a caller whose own stream is already restricted to a subset of the CUs has partitioned the chip itself: its calls are not split again
static bool stream_is_cu_masked(hipStream_t st) {
    const int ncu = device_cus();
    uint32_t mask[16] = {0};
    const int words = std::min(16, (ncu + 31) / 32);
    if (hipExtStreamGetCUMask(st, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); return false; }
    int bits = 0;
    for (int w = 0; w < words; ++w) bits += __builtin_popcount(mask[w]);
    return bits > 0 && bits < ncu;
}

static bool split_applies(int batch, int64_t m, int64_t n) {
    const int mode = g_split_mode.load(std::memory_order_relaxed);
    if (mode == 0) return false;
    if (mode < 0) {
        const char* e = getenv("ASVD_SPLIT");   // read per call: tests and A/B runs toggle it inside one process
        if (e && atoi(e) == 0) return false;
    }
    return batch >= 4 && std::min(m, n) >= 3072;   // measured: 4 x 4096^2 -3 %, 8: -8 %, 12: -6 %, 16: -9 %, 32: -6 %; 2048 columns: nothing
}

void asvd_svd_set_split(int mode) { g_split_mode.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }

int asvd_svd_worksize(int batch, int64_t m, int64_t n, int want_vectors, size_t* bytes) {
    if (!bytes) return ASVD_E_BADARG;
    size_t need = 0;
    int rc = worksize_one(batch, m, n, want_vectors, &need);
    if (rc) return rc;
    if (split_applies(batch, m, n)) {   // room for the two halves, each planned for half of the CUs
        const int saved = g_call_cus;
        g_call_cus = std::max(1, device_cus() / 2);
        size_t h0 = 0, h1 = 0;
        rc = worksize_one((batch + 1) / 2, m, n, want_vectors, &h0);
        if (!rc) rc = worksize_one(batch / 2, m, n, want_vectors, &h1);
        g_call_cus = saved;
        if (rc) return rc;
        need = std::max(need, ((h0 + 255) & ~(size_t)255) + h1);
    }
    *bytes = need;
    return ASVD_OK;
}

// allow_f16: the fused update + Gram kernel of the dense sweeps (split-fp16 arithmetic with power-of-two column scales taken from the carried
// column norms, twolevel.h).  *retry_plain is set, and nothing is finalised, when a problem turned NaN on that path: the caller then repeats the
// call with the separate passes (fp32 Gram pass, split-bf16 update pass), which need no scales — a NaN that is in the data comes back as NaN.
static int svd_direct_run(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                          const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                          float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                          int* info_host, void* stream, bool manage_profile, bool allow_f16, bool* retry_plain) {
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
    sc.pair_order = pair_order_xor() ? 1 : 0;
    sc.super_order = super_grouped_for(p) ? 2 : 1;
    if (sc.super_order == 2) set_group_table(sc, p.ns, group_bits_for(p.ns));
    const int nsteps = pair_order_xor() ? 2 * p.npairs - 1 : p.nb - 1;
    // panels whose convergence is enforced: those holding the k leading columns, plus one panel of margin
    const int kb = (int)(ceil_div64(k, PB) + 1 < p.nb ? ceil_div64(k, PB) + 1 : p.nb);
    // one inner sweep of the 64x64 eigen-solve per visit: with the XOR schedule a second one no longer saves outer sweeps
    // (measured 26.3 vs 25.9 SVD/s; under the round-robin order two inner sweeps cut 15 -> 13 outer sweeps)
    constexpr int inner_sweeps = 1;
    int sweep = 0;
    // Step schedule of a single-level dense sweep (XOR distances d, stored as step = d-1).  The local levels d = 1..L are run twice at the
    // start of every sweep: the strongest couplings of the sorted, preconditioned matrix sit between neighbouring panels, and a second pass
    // over them is cheap (L extra steps of P-1) — docs/history.md (round 1).  L = min(7, P/16 - 1): measured at 4096^2 (P = 128) 8 -> 7 sweeps, +2.4 %.
    std::vector<int> sched;
    {
        const int P2 = nsteps + 1;
        const int L = pair_order_xor() ? std::min(std::max(0, std::min(7, P2 / 16 - 1)), nsteps) : 0;
        for (int d = 1; d <= L; ++d) sched.push_back(d - 1);
        for (int st2 = 0; st2 < nsteps; ++st2) sched.push_back(st2);
    }
    // ---- sparse-sweep state (see fullcheck_kernel) ----
    unsigned char* pflag = (unsigned char*)(wb + p.off_pflag);
    int* plist_dev = (int*)(wb + p.off_plist);
    float* dnorm = (float*)(wb + p.off_ina);  // squared column norms of the snapshot (finalize overwrites this buffer later)
    const bool sparse_allowed = pair_order_xor() && p.nb >= 8 && !(getenv("ASVD_SPARSE") && atoi(getenv("ASVD_SPARSE")) == 0);
    constexpr double sparse_frac = 0.5;
    bool sparse = false;
    std::vector<unsigned char> hflag;
    std::vector<int> hlist;
    std::vector<int> sl_off((size_t)(nsteps > 0 ? nsteps : 1), 0), sl_cnt((size_t)(nsteps > 0 ? nsteps : 1), 0);
    bool fused_used = false;
    int* hist_dev = nullptr;
    if (getenv("ASVD_DEBUG_HIST")) { ASVD_HIP_CHECK(hipMalloc(&hist_dev, 10 * sizeof(int))); }
    // every exit of this function — the ASVD_HIP_CHECK returns included — frees the debug histogram and closes the profile it opened
    struct ExitGuard {
        int*& hist; bool prof;
        ~ExitGuard() { if (hist) { (void)hipFree(hist); hist = nullptr; } if (prof) prof_end(); }
    } exit_guard{hist_dev, g_prof_enabled && manage_profile};
    // buffers of the two-level sweeps as the kernels see them
    EvdV3 v3{};
    v3.ns = p.ns;
    v3.nbpan = p.nb;
    v3.Gd32 = (float*)(wb + p.off_gd32);
    v3.Q0 = (float*)(wb + p.off_q0);
    v3.D0 = (float*)(wb + p.off_d0);
    v3.Qfin = (float*)(wb + p.off_qfin);
    v3.subact = (int*)(wb + p.off_subact);
    v3.Din = (float*)(wb + p.off_din);
    v3.hist = hist_dev;
    float* Gx6 = (float*)(wb + p.off_gx6);
    for (; sweep < max_sweeps; ++sweep) {
        const auto sweep_t0 = std::chrono::steady_clock::now();
        if (hist_dev) ASVD_HIP_CHECK(hipMemsetAsync(hist_dev, 0, 10 * sizeof(int), st));
        ASVD_HIP_CHECK(hipMemsetAsync(maxoff, 0, (size_t)batch * sizeof(int), st));
        ASVD_HIP_CHECK(hipMemsetAsync(nrot, 0, (size_t)batch * sizeof(int), st));
        ASVD_HIP_CHECK(hipMemsetAsync(nupd, 0, (size_t)batch * sizeof(int), st));
        long long marked_total = 0;
        if (sparse) {
            // 1. snapshot of all couplings
            {
                ProfScope ps(5, st);
                ASVD_HIP_CHECK(hipMemsetAsync(pflag, 0, (size_t)batch * p.nb * p.nb, st));
                const unsigned nt = (unsigned)ceil_div64(p.nb, 4);
                const char* es8 = getenv("ASVD_SNAP_I8");   // read per call
                const bool snap8 = p.nb >= 8 && p.m_pad <= 32768 && !(es8 && atoi(es8) == 0);
                if (snap8) panel_sumsq_max_kernel<<<dim3(p.nb, batch), 256, 0, st>>>(X, p.panel_stride, p.batch_stride, p.m_pad, p.n_pad, dnorm, (int*)(wb + p.off_snapex), done);
                else panel_sumsq_kernel<<<dim3(p.nb, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.m_pad, p.n_pad, dnorm, done);
                if (snap8) {
                    // X^T X from int8 digit planes (snapshot_i8.h): digitise once, eight exact digit products per fp32 product
                    int* sex = (int*)(wb + p.off_snapex);
                    signed char* spl = (signed char*)(wb + p.off_snap);
                    const int kgs = (int)(round_up64(p.m_pad, 64) / 16);
                    const int64_t plane_stride = (int64_t)p.nb * kgs * 512;
                    split_i8_kernel<<<dim3(p.nb, (unsigned)ceil_div64(kgs, 8), batch), 256, 0, st>>>(X, p.panel_stride, p.batch_stride, p.nb, p.m_pad, p.n_pad, sex, 0,
                                                                                                   kgs, spl, plane_stride, nullptr, done);
                    ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)fullcheck_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GI_STAGE_BYTES));
                    fullcheck_i8_kernel<<<dim3(nt * (nt + 1) / 2, batch), 512, 2 * GI_STAGE_BYTES, st>>>(spl, plane_stride, p.nb, kgs, p.n_pad, sex, dnorm, tol, kb,
                                                                                                      pflag, maxoff, done, (int)nt);
                } else {
                    fullcheck_kernel<<<dim3(nt, nt, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.nb, p.m_pad, p.n_pad, dnorm, tol, kb, pflag,
                                                                          maxoff, done);
                }
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
            // (The list slots are indexed by `step` below.)
            hlist.clear();
            std::vector<std::vector<int>> tmp((size_t)batch * nsteps);  // [b * nsteps + round]
            const int words = (p.nb + 63) / 64;
            std::vector<uint64_t> used;  // [round][words] of the problem being packed
            for (int b = 0; b < batch; ++b) {
                if (host_done[b]) continue;
                const unsigned char* f = hflag.data() + (size_t)b * p.nb * p.nb;
                bool packed = true;
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
                        tmp[(size_t)b * nsteps + r].push_back((I << 16) | J);
                        ++cnt;
                    }
                if (packed) marked_total += cnt;
                else {  // one list per XOR step
                    for (int r = 0; r < nsteps; ++r) tmp[(size_t)b * nsteps + r].clear();
                    for (int I = 0; I < p.nb; ++I)
                        for (int J = I + 1; J < p.nb; ++J)
                            if (f[(size_t)I * p.nb + J]) { tmp[(size_t)b * nsteps + ((I ^ J) - 1)].push_back((I << 16) | J); ++marked_total; }
                }
            }
            for (int step = 0; step < nsteps; ++step) {
                size_t mx = 0;
                for (int b = 0; b < batch; ++b) mx = std::max(mx, tmp[(size_t)b * nsteps + step].size());
                sl_cnt[step] = (int)mx;
                sl_off[step] = (int)hlist.size();
                for (int b = 0; b < batch; ++b) {
                    const auto& v = tmp[(size_t)b * nsteps + step];
                    for (size_t i = 0; i < mx; ++i) hlist.push_back(i < v.size() ? v[i] : -1);
                }
            }
            if (debug) fprintf(stderr, "[asvd_svd]   lists built at %.2f ms (%zu slots)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sweep_t0).count(), hlist.size());
            if (!hlist.empty()) ASVD_HIP_CHECK(hipMemcpyAsync(plist_dev, hlist.data(), hlist.size() * sizeof(int), hipMemcpyHostToDevice, st));
        }
        // two-level dense sweep: only the internal step d = 1 runs through the single-level kernels (it also refreshes the carried
        // diagonal blocks); the super-steps follow below
        const bool two_now = p.two && !sparse;
        const int nsched = sparse ? nsteps : (two_now ? 1 : (int)sched.size());
        for (int si = 0; si < nsched; ++si) {
            const int step = sparse ? si : (two_now ? si : sched[si]);
            if (sparse) {
                const int slots = sl_cnt[step];
                if (slots == 0) continue;  // no marked pair meets in this round
                const int* pl = plist_dev + sl_off[step];
                // few pairs per launch: split the rows further to fill the CUs, within the partial-Gram capacity of a problem
                const int64_t chunks = p.m_pad / 32, cap = (int64_t)p.npairs * p.nsplit / slots;
                int64_t ns = std::max<int64_t>(p.nsplit, ceil_div64(3 * call_cus(), (int64_t)slots * batch));
                ns = std::min<int64_t>(std::min<int64_t>(ns, cap), std::min<int64_t>(chunks, 64));
                if (ns < 1) ns = 1;
                const int rps = (int)(ceil_div64(chunks, ns) * 32);
                const int nsp = (int)ceil_div64(p.m_pad, rps);
                const int64_t iters = ceil_div64(p.R_upd, 128);
                int64_t nc = std::min<int64_t>(iters, std::max<int64_t>(1, ceil_div64(4 * call_cus(), (int64_t)slots * batch)));
                const int rpw = (int)(ceil_div64(iters, nc) * 128);
                const int nch = (int)ceil_div64(p.R_upd, rpw);
                {
                    ProfScope ps(6, st);
                    gram_kernel<<<dim3(nsp, slots, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.nb, step, p.m_pad, rps, Gpart, done, pl, slots);
                }
                {
                    ProfScope ps(2, st);
                    EvdV3 none{};
                    none.hist = hist_dev;
                    launch_evdw0(false, slots, batch, st, sc, Gpart, nsp, Qbuf, active, maxoff, nrot, done, tol, inner_sweeps, p.nb, step, kb, pl, slots, none);
                }
                {
                    ProfScope ps(7, st);
                    update_kernel<<<dim3(nch, slots, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.nb, step, p.R_upd, rpw, Qbuf, active, done, pl, slots);
                }
                continue;
            }
            {
                ProfScope ps(6, st);
                gram_kernel<<<dim3(p.nsplit, p.npairs, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.nb, step, p.m_pad, p.rows_per_split, Gpart,
                                                                          done, nullptr, 0);
            }
            {
                ProfScope ps(2, st);
                if (two_now) {  // internal step: also emits the fresh carried diagonal blocks of both panels of every pair
                    launch_evdw0(true, p.npairs, batch, st, sc, Gpart, p.nsplit, Qbuf, active, maxoff, nrot, done, tol, inner_sweeps, p.nb, step, kb, nullptr, 0, v3);
                } else {
                    EvdV3 none{};
                    none.hist = hist_dev;
                    launch_evdw0(false, p.npairs, batch, st, sc, Gpart, p.nsplit, Qbuf, active, maxoff, nrot, done, tol, inner_sweeps, p.nb, step, kb, nullptr, 0, none);
                }
            }
            {
                ProfScope ps(7, st);
                update_kernel<<<dim3(p.nchunks, p.npairs, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.nb, step, p.R_upd, p.rows_per_wg, Qbuf, active,
                                                                             done, nullptr, 0);
            }
        }
        if (two_now) {
            const bool super_grp = super_grouped_for(p);
            const int sgb = super_grp ? group_bits_for(p.ns) : 0, sG1 = (1 << sgb) - 1;   // grouped schedule: group size - 1
            const int nsuper = super_grp ? grouped_steps(p.ns, sgb) : 2 * p.npairs_s - 1;
            // supgram: the update of step D also leaves the Gram tiles of the step that follows (one pass instead of two); the stand-alone Gram
            // pass then runs only in front of the first super-step.  ASVD_SUPGRAM=0 keeps the separate passes (fp32 Gram pass, split-bf16 update
            // pass: also what a call falls back to when the split-fp16 path of the fused kernel turns a problem NaN).
            const bool fuse_ug = allow_f16 && p.npairs_s >= 2 && !(getenv("ASVD_SUPGRAM") && atoi(getenv("ASVD_SUPGRAM")) == 0);
            fused_used = fused_used || fuse_ug;
            if (fuse_ug)
                ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)supgram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SUPGRAM_SMEM_FLOATS * sizeof(float))));
            // on a padded XOR schedule (ns not a power of two and no grouped schedule) many super-steps hold few real pairs and the fused kernel,
            // which reads every panel of a quad with a present member, loses to the separate passes, which touch only the pairs that exist
            // (13B shapes, ns = 80 in a 128-wide schedule: always fused 21.4 s, never fused 16.9 s for the model): fused on full schedules only
            const bool ns_pow2 = (p.ns & (p.ns - 1)) == 0;
            // ASVD_SPREAD_FROM=<k> (measurement knob, XOR schedules over 4^j super-panels only): from dense sweep k (0-based) on, the XOR
            // distances run in LINE-SPREAD order — triples {d, w(d), w^2(d)}, w: base-4 digits 1 -> 2 -> 3 -> 1, which XOR to zero, so that the
            // four super-panels {a, a^D, a^E, a^D^E} of a quad are closed under all three steps of a triple (what a
            // three-steps-per-pass kernel would need).  Any order of the distances is a valid sweep; this one gives up nearest-neighbour-first.
            std::vector<int> dist(nsuper);
            for (int di = 0; di < nsuper; ++di) dist[di] = di + 1;
            {
                const char* es = getenv("ASVD_SPREAD_FROM");
                int lg = 0;
                while ((1 << lg) < p.ns) ++lg;
                if (es && !super_grp && ns_pow2 && (lg % 2) == 0 && sweep >= atoi(es)) {
                    auto w = [](int d) { int o = 0; for (int sh = 0; sh < 30; sh += 2) { const int g = (d >> sh) & 3; o |= (g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 3 : 1))) << sh; } return o; };
                    std::vector<char> used(p.ns, 0);
                    int at = 0;
                    for (int d = 1; d < p.ns; ++d) {
                        if (used[d]) continue;
                        const int d2 = w(d), d3 = w(d2);
                        used[d] = used[d2] = used[d3] = 1;
                        dist[at++] = d; dist[at++] = d2; dist[at++] = d3;
                    }
                }
            }
            for (int di = 0; di < nsuper; ++di) {
                const int D = dist[di];
                const int E = (di + 1 < nsuper) ? dist[di + 1] : 0;  // 0: last super-step of the sweep
                auto fused_after = [&](int dj) {  // does the launch of super-step index dj also leave the tiles of index dj + 1 ?
                    if (!fuse_ug || dj < 0 || dj + 1 >= nsuper) return false;
                    if (super_grp) {  // consecutive steps inside the groups (XOR distances 1..15), or consecutive offsets of the same round of group pairs
                        const int s0 = dj, s1 = dj + 1;   // 0-based super-steps
                        return s1 < sG1 || (s0 >= sG1 && ((s0 - sG1) >> sgb) == ((s1 - sG1) >> sgb));
                    }
                    return ns_pow2;
                };
                const bool gram_in = !fused_after(di - 1);   // tiles of this step not left by the previous launch
                const bool gram_out = fused_after(di);
                const int gx_slots = std::max(p.nsplit_s, p.nchunks_q);  // partial-tile slots per super-pair in the buffer
                (void)gx_slots;
                v3.nsplit6 = gram_in ? p.nsplit_s : p.nchunks_q;
                v3.Gx6 = Gx6;
                if (gram_in) {
                    ProfScope ps(1, st);
                    sgram6_kernel<<<dim3(p.nsplit_s, p.npairs_s, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.ns, D, p.m_pad, p.rows_per_split_s, Gx6, done);
                }
                {
                    ProfScope ps(2, st);  // both inner steps of every super-pair in one launch, one wave per 64x64 solve (evd_wave.hip)
                    launch_evdw12(p.npairs_s, batch, st, sc, maxoff, nrot, done, tol, inner_sweeps, p.nb, D - 1, kb, v3, sweep, p.cols >= 2048 ? 2 : 0);
                }
                {
                    ProfScope ps(gram_out ? 8 : 3, st);
                    if (gram_out)
                        supgram_kernel<<<dim3(p.nchunks_q, (2 * p.npairs_s) / 4, batch), 512, SUPGRAM_SMEM_FLOATS * sizeof(float), st>>>(
                            sc, X, p.panel_stride, p.batch_stride, p.ns, D, E, p.R_upd, p.m_pad, p.rows_per_wg_q, v3.Qfin, v3.subact, v3.Din, Gx6, done, nupd, p.npairs_s);
                    else
                        supdate_split_kernel<<<dim3(p.nchunks_s, p.npairs_s, batch), 256, 0, st>>>(sc, X, p.panel_stride, p.batch_stride, p.ns, D, p.R_upd,
                                                                                                p.rows_per_wg_s, v3.Qfin, v3.subact, done, nupd);
                }
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

    if (fused_used && retry_plain) {
        for (int b = 0; b < batch; ++b)
            if (status[b] == ASVD_N_NAN) {
                if (debug) fprintf(stderr, "[asvd_svd] problem %d turned NaN on the split-fp16 path: repeating the call with the separate passes\n", b);
                *retry_plain = true;
                return ASVD_OK;
            }
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

static int svd_direct(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                      const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                      float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                      int* info_host, void* stream, bool manage_profile) {
    bool retry = false;
    int rc = svd_direct_run(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes, info_host,
                            stream, manage_profile, true, &retry);
    if (rc >= 0 && retry) g_last_path |= ASVD_PATH_PLAIN_RETRY;
    if (rc >= 0 && retry)
        rc = svd_direct_run(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes, info_host,
                            stream, manage_profile, false, nullptr);
    return rc;
}


// ---------------------------------------------------------------------------------------------------------------------
// tall path driver (see the kernel block "Tall problems" above)
static bool tall_wanted(const Plan& p) {
    // The reduction pays for every shape: for tall problems it shrinks each Jacobi step from rows x cols to cols x cols, and for all
    // of them the norm-sorted Cholesky-QR is a preconditioner (Jacobi on R^T: 14 -> 10 sweeps at 4096^2, better orthogonality).
    if (getenv("ASVD_NO_REDUCE")) return false;
    return p.cols >= 128;
}

static bool nn_i8_wanted() {
    const char* e = getenv("ASVD_NN_I8");   // read per call
    return !(e && atoi(e) == 0);
}
static bool gram_i8_wanted() {
    const char* e = getenv("ASVD_GRAM_I8");   // read per call: tests and A/B runs toggle it inside one process
    return !(e && atoi(e) == 0);
}

// G (upper 32-blocks) = X^T X of the packed panels through the int8 digit planes (gram_i8.h).  scratch: >= 3 * n_pad * 64 * batch bytes; the
// rows go in segments of what fits (and of at most 32768 rows: the int32 accumulators), each added to G in fp64.  ex: n_pad column exponents per
// problem (colmaxexp_kernel) in the order of X.  inv / exs / dp (all or none): column q of X goes to plane position inv[q], exs / dp are the exponents and
// norms in plane order, and G_ij / (dp_i dp_j) is stored — the sorted, unit-scaled Gram matrix the Cholesky factorisation starts from, without a pass
// over an unsorted one.
static int launch_gram_i8(const float* Xp, int64_t panel_stride, int64_t batch_stride, int nb, int m_pad, int n_pad, int batch, double* G,
                          int64_t ldg, int64_t gbs, signed char* scratch, size_t scratch_bytes, const int* ex, const int* inv, const int* exs,
                          const double* dp, int seg_rows_max, hipStream_t st) {
    int64_t cap_rows = (int64_t)(scratch_bytes / ((size_t)3 * n_pad * batch)) / 64 * 64;
    if (cap_rows > 32768) cap_rows = 32768;
    if (seg_rows_max >= 64 && cap_rows > seg_rows_max / 64 * 64) cap_rows = seg_rows_max / 64 * 64;
    if (cap_rows < 64) return ASVD_E_WORKSPACE;
    const int64_t m64 = round_up64(m_pad, 64);
    const int nseg = (int)ceil_div64(m64, cap_rows);
    const int seg = (int)round_up64(ceil_div64(m64, nseg), 64);
    ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)gram_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GI_STAGE_BYTES));
    const int nt = (nb + 3) / 4, ntri = nt * (nt + 1) / 2;
    const int order = getenv("ASVD_GI_ORDER") ? atoi(getenv("ASVD_GI_ORDER")) : 0;   // measurement knob (block order of gram_i8_kernel)
    for (int s = 0; s < nseg; ++s) {
        const int r0 = s * seg;
        const int rows = (int)std::min<int64_t>(seg, m64 - r0);
        if (rows <= 0) break;
        const int kgs = rows / 16;
        const int64_t plane_stride = (int64_t)nb * kgs * 512;
        split_i8_kernel<<<dim3(nb, (unsigned)ceil_div64(kgs, 8), batch), 256, 0, st>>>(Xp, panel_stride, batch_stride, nb, m_pad, n_pad, ex, r0, kgs,
                                                                                       scratch, plane_stride, inv, nullptr);
        gram_i8_kernel<<<dim3(ntri, batch), 512, 2 * GI_STAGE_BYTES, st>>>(scratch, plane_stride, nb, kgs, n_pad, inv ? exs : ex, G, ldg, gbs, s > 0 ? 1 : 0, nt,
                                                                           dp, order);
    }
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

struct TallLayout {
    size_t off_xp, off_g, off_gs, off_dp, off_df, off_perm, off_dg, off_d, off_fail, off_r, off_vr, off_part, off_inv, off_bex, off_aex, off_vmax, off_gex, off_ex, off_exs, off_cinv, off_inner, inner_bytes, total;
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
    // column exponents of the int8 Gram matrix (gram_i8.h), original and sorted order
    t.off_ex = take((size_t)p.n_pad * batch * sizeof(int));
    t.off_exs = take((size_t)p.n_pad * batch * sizeof(int));
    t.off_cinv = take((size_t)p.n_pad * batch * sizeof(int));   // plane position of every column (inverse of the norm sort)
    // exponents of the int8 long-side product (nn_gemm_i8.h): columns of X, rows of X, columns of Vr (key of the maximum, then the exponent)
    t.off_bex = take(want_vectors ? (size_t)p.n_pad * batch * sizeof(int) : 0);
    t.off_aex = take(want_vectors ? (size_t)round_up64(p.m_pad, 128) * batch * sizeof(int) : 0);
    t.off_vmax = take(want_vectors ? (size_t)round_up64(k, 32) * batch * sizeof(unsigned) : 0);
    t.off_gex = take(want_vectors ? (size_t)round_up64(k, 32) * batch * sizeof(int) : 0);
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
    int* fail = (int*)(wb + t.off_fail);
    float* R = (float*)(wb + t.off_r);
    float* Vr = (float*)(wb + t.off_vr);
    const int64_t ldg = p.n_pad, gbs = (int64_t)p.n_pad * p.n_pad;
    const int nbk = p.n_pad / CB;
    // Jacobi runs on R^T: the leading right vectors of X are then the directly rotated columns (orthogonal to 1e-6) instead of
    // backsolved ones, which matters because the long-side vectors X v / sigma amplify the error of v by sigma_1 / sigma_j;
    // row-scaled problems (wide layers: the activation scales sit on the long side) also need 3 fewer sweeps.

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
        // G = X^T X: exact integer arithmetic on the int8 matrix pipe (gram_i8.h), or the fp64 matrix instructions (ASVD_GRAM_I8=0), then
        // sort columns by decreasing norm (stable, padding last; 14 -> 10 sweeps, unsorted saves nothing), permute + unit-scale the Gram matrix.
        // The int8 path takes the norms from the data first (fp64, colmaxexp_kernel) and stores the sorted, scaled matrix directly: Gs = D^-1 P^T G~ P D^-1
        // with G~ the exact Gram matrix of the digitised columns and D the norms of the original ones — diagonal 1 + O(1e-7) instead of exactly 1,
        // which is a column scaling like any other (R is un-scaled with the same D).  Digit planes in the G region, which this path does not use.
        // A Cholesky breakdown on the int8 path is retried ONCE with the fp64 Gram matrix before the call gives up on the reduction: the digitised columns
        // differ from X by ~2^-25 of their largest entry, which decides the sign of a pivot only where singular values sit at the fp32 rounding level of
        // X itself (rank n/4 + 1e-4 noise under abs_mean scaling: `profiles/r6_families.txt`) — there the exact Gram matrix of the exact X still factors.
    }
    std::vector<int> hfail(batch, 0);
    bool use_i8 = gram_i8_wanted();
    for (int attempt = 0; attempt < 2; ++attempt) {
        {
        ProfScope ps(0, st);
        if (attempt) ASVD_HIP_CHECK(hipMemsetAsync(fail, 0, (size_t)batch * sizeof(int), st));
        if (use_i8) {
            int* ex = (int*)(wb + t.off_ex);
            int* exs = (int*)(wb + t.off_exs);
            colmaxexp_kernel<<<dim3(p.nb, batch), 256, 0, st>>>(Xp, p.panel_stride, p.batch_stride, p.m_pad, p.n_pad, ex, d);
            d_to_float_kernel<<<(unsigned)ceil_div64((int64_t)p.n_pad * batch, 256), 256, 0, st>>>(d, p.n_pad * batch, dF);
            rank_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(dF, p.n_pad, cperm);
            int* cinv = (int*)(wb + t.off_cinv);
            perm_gather_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(ex, d, cperm, p.n_pad, exs, dp, cinv);
            rc = launch_gram_i8(Xp, p.panel_stride, p.batch_stride, p.nb, p.m_pad, p.n_pad, batch, Gs, ldg, gbs, (signed char*)G,
                                (size_t)gbs * batch * sizeof(double), ex, cinv, exs, dp, 0, st);
            if (rc) return rc;
        } else {
            gram64_kernel<<<dim3(p.nb, (unsigned)ceil_div64(p.nb, 4), batch), 256, 0, st>>>(Xp, p.panel_stride, p.batch_stride, p.nb, p.m_pad, G, ldg, gbs);
            chol_diag_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(G, ldg, gbs, p.n_pad, d);
            d_to_float_kernel<<<(unsigned)ceil_div64((int64_t)p.n_pad * batch, 256), 256, 0, st>>>(d, p.n_pad * batch, dF);
            rank_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), batch), 256, 0, st>>>(dF, p.n_pad, cperm);
            g_permute_scale_kernel<<<dim3((unsigned)ceil_div64(p.n_pad, 256), p.n_pad, batch), 256, 0, st>>>(G, ldg, gbs, d, cperm, p.n_pad, Gs, dp);
        }
        // block rows in groups of six (four until round 6): inside a group each finished row updates the rest of the group's strip (K = 64), the matrix
        // behind the group is updated once per group (K = 384)
        // (Round 6, measured and removed — profiles/r6_chol_lookahead.txt: the chain of group g + 1 on a second, equally CU-masked stream beside the bulk
        // of group g's trailing update, the next group's strip updated first.  Reduction 48.1-48.5 -> 52.3 ms per 32 x 4096^2: the two extra launches and
        // event hand-overs per group cost more than the 0.5 ms chain they hide, and the trailing update loses the CUs the chain occupies.)
        // (block rows per group, 32 x 4096^2, reduction ms: 2: 48.0 · 4: 44.4-44.6 · 5: 42.1 · 6: 42.2-42.6 · 7: 41.9 · 8: 44.9-45.2 — profiles/r6_chol_group.txt)
        int cg = 6;
        if (const char* ecg = getenv("ASVD_CHOL_GROUP")) cg = std::max(1, std::min(16, atoi(ecg)));   // measurement knob: block rows per grouped trailing update
        for (int j0 = 0; j0 < nbk; j0 += cg) {
            const int j1 = std::min(nbk, j0 + cg);
            for (int jb = j0; jb < j1; ++jb) {
                chol_diag_wave_kernel<<<batch, 64, 0, st>>>(Gs, ldg, gbs, jb, fail, Dg, nbk);
                if (jb + 1 < nbk) chol_trsm_kernel<<<dim3(nbk - jb - 1, batch), 256, 0, st>>>(Gs, ldg, gbs, jb);
                if (jb + 1 < j1) chol_syrk_kernel<<<dim3(j1 - jb - 1, nbk - jb - 1, batch), 256, 0, st>>>(Gs, ldg, gbs, jb, nbk);
            }
            if (j1 < nbk) chol_syrk_multi_kernel<<<dim3(nbk - j1, nbk - j1, batch), 256, 0, st>>>(Gs, ldg, gbs, j0, j1 - j0, nbk);
        }
        r_to_f32_t_kernel<<<dim3((unsigned)(p.n_pad / 32), (unsigned)(p.n_pad / 32), batch), 256, 0, st>>>(Gs, ldg, gbs, Dg, dp, p.n_pad, R, gbs);  // n_pad is a multiple of 64
        }
        ASVD_HIP_CHECK(hipMemcpyAsync(hfail.data(), fail, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, st));
        ASVD_HIP_CHECK(hipStreamSynchronize(st));
        bool broke = false;
        for (int b = 0; b < batch; ++b) broke = broke || hfail[b];
        if (!broke) break;
        if (use_i8 && attempt == 0) {
            if (getenv("ASVD_DEBUG")) fprintf(stderr, "[asvd_svd] Cholesky breakdown on the int8 Gram matrix: once more with the fp64 Gram matrix\n");
            use_i8 = false;
            g_last_path |= ASVD_PATH_GRAM_RETRY;
            continue;
        }
        break;
    }
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
    float* const* inner_U = want_vectors ? vperm.data() : nullptr;   // left vectors of R^T = right vectors of X
    float* const* inner_V = nullptr;
    rc = svd_direct(batch, rp.data(), ASVD_F32, p.cols, p.cols, p.n_pad, nullptr, 0, inner_U, S_host, inner_V, k, max_sweeps, tol,
                    wb + t.off_inner, t.inner_bytes, info_host, stream, false);
    if (rc < 0) return rc;
    if (want_vectors) {
        ProfScope ps(4, st);
        // epilogues (un-permute, long-side product, sigma refinement), up to TALL_ZB problems per launch
        const int nsp = (int)std::min<int64_t>(64, ceil_div64(p.rows, 256));
        const int rps = (int)ceil_div64(p.rows, nsp);
        for (int b0 = 0; b0 < batch; b0 += TALL_ZB) {
            const int zb = std::min(TALL_ZB, batch - b0);
            TallBatch tb{};
            bool any_long = false;
            for (int z = 0; z < zb; ++z) {
                const int b = b0 + z;
                tb.vperm[z] = vperm[b];
                tb.vr[z] = vr[b];
                tb.lng[z] = p.transposed ? (V_host ? V_host[b] : nullptr) : (U_host ? U_host[b] : nullptr);
                tb.S[z] = S_host[b];
                any_long = any_long || tb.lng[z];
            }
            row_unpermute_kernel<<<dim3((unsigned)ceil_div64(k, 256), p.cols, zb), 256, 0, st>>>(tb, cperm + (int64_t)b0 * p.n_pad, p.n_pad, p.cols, (int)k);
            if (!any_long) continue;
            if (nn_i8_wanted() && p.cols <= 32768) {
                // Y = X Vr on the int8 matrix pipe (nn_gemm_i8.h).  Digit planes: Vr in the G region, X^T (row segments of what fits) in the Gs region —
                // both free since R left them; exponents in their own small arrays.
                const float* Xz = Xp + (int64_t)b0 * p.batch_stride;
                int* bex = (int*)(wb + t.off_bex) + (int64_t)b0 * p.n_pad;
                const int rows_pad = (int)round_up64(p.m_pad, 128), kp = (int)round_up64(k, 32);
                int* aex = (int*)(wb + t.off_aex) + (int64_t)b0 * rows_pad;
                unsigned* vmax = (unsigned*)(wb + t.off_vmax) + (int64_t)b0 * kp;
                int* gex = (int*)(wb + t.off_gex) + (int64_t)b0 * kp;
                const int cgs = p.n_pad / 16, jps = kp / 32;
                const int64_t strideB = (int64_t)jps * cgs * 512;
                signed char* planesB = (signed char*)G;
                signed char* planesA = (signed char*)Gs;
                colexp_from_norm_kernel<<<(unsigned)ceil_div64((int64_t)p.n_pad * zb, 256), 256, 0, st>>>(d + (int64_t)b0 * p.n_pad, p.n_pad * zb, bex);
                rowmaxexp_kernel<<<dim3((unsigned)(rows_pad / 8), zb), 256, 0, st>>>(Xz, p.panel_stride, p.batch_stride, p.nb, p.rows, rows_pad, bex, p.n_pad, aex);
                ASVD_HIP_CHECK(hipMemsetAsync(vmax, 0, (size_t)kp * zb * sizeof(unsigned), st));
                const int vchunks = (int)std::min<int64_t>(32, ceil_div64(p.cols, 64));
                const int vrpc = (int)ceil_div64(p.cols, vchunks);
                vcolmax_kernel<<<dim3((unsigned)ceil_div64(k, 256), vchunks, zb), 256, 0, st>>>(tb, p.cols, (int)k, k, bex, p.n_pad, vrpc, vmax, kp);
                split_v_i8_kernel<<<dim3(jps, (unsigned)ceil_div64(cgs, 8), zb), 256, 0, st>>>(tb, p.cols, (int)k, k, bex, p.n_pad, vmax, kp, cgs, planesB, strideB, gex);
                ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)nn_gemm_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GI_STAGE_BYTES));
                int64_t seg = ((int64_t)8 * p.n_pad / 3) / 128 * 128;   // rows of X^T planes per problem that fit the Gs region
                if (seg > rows_pad) seg = rows_pad;
                for (int64_t r0 = 0; r0 < rows_pad; r0 += seg) {
                    const int rps = (int)(std::min<int64_t>(seg, rows_pad - r0) / 32);
                    const int64_t strideA = (int64_t)rps * cgs * 512;
                    split_xt_i8_kernel<<<dim3(rps, (unsigned)ceil_div64(cgs, 8), zb), 256, 0, st>>>(Xz, p.panel_stride, p.batch_stride, p.rows, rows_pad, bex, p.n_pad,
                                                                                                  aex, (int)r0, cgs, planesA, strideA);
                    const int gx = (int)ceil_div64(jps, 4), gy = (int)ceil_div64(rps, 4);
                    nn_gemm_i8_kernel<<<dim3(gx * gy, zb), 512, 2 * GI_STAGE_BYTES, st>>>(tb, planesA, strideA, rps, planesB, strideB, jps, cgs, aex, rows_pad, gex, kp,
                                                                                         (int)r0, p.rows, (int)k, k, gx, gy,
                                                                                         getenv("ASVD_NI_ORDER") ? atoi(getenv("ASVD_NI_ORDER")) : 1);
                }
            } else {
                nn_gemm_split_kernel<<<dim3((unsigned)ceil_div64(k, 128), (unsigned)ceil_div64(p.rows, 128), zb), 256, 0, st>>>(
                    tb, Xp + (int64_t)b0 * p.batch_stride, p.panel_stride, p.batch_stride, p.nb, p.rows, p.cols, k, (int)k, k);
            }
            // sigma_j = |X v_j| and unit left vectors
            double* part = (double*)(wb + t.off_part) + (size_t)b0 * 64 * k;
            float* invs = (float*)(wb + t.off_inv) + (size_t)b0 * k;
            colsumsq_kernel<<<dim3((unsigned)ceil_div64(k, 64), nsp, zb), 256, 0, st>>>(tb, k, p.rows, (int)k, rps, part, (int64_t)64 * k);
            colfinish_kernel<<<zb, 256, 0, st>>>(tb, part, (int64_t)64 * k, nsp, (int)k, invs);
            colscale_kernel<<<dim3((unsigned)ceil_div64(k, 256), (unsigned)ceil_div64(p.rows, 32), zb), 256, 0, st>>>(tb, k, p.rows, (int)k, invs);
        }
        ASVD_HIP_CHECK(hipStreamSynchronize(st));
        ASVD_HIP_CHECK(hipGetLastError());
    }
    return rc;
}

// Test hook (tests/test_gpu_twolevel.py): ONE launch of the two-level update kernel (split-bf16) on caller-built panels, so that the kernel
// can be checked against a plain fp64 product.  X: [batch][nb][R][32] fp32 panels; Qfin: [batch][npairs][128*128]; subact: [batch][npairs][4];
// done, nupd: [batch] ints (all device pointers).
int asvd_test_supdate(float* X, int64_t panel_stride, int64_t batch_stride, int ns, int D, int R, int rows_per_wg, const float* Qfin,
                      const int* subact, const int* done, int* nupd, int nchunks, int npairs, int batch, void* stream) {
    if (!X || !Qfin || !subact || !done || !nupd || ns < 2 || D < 1 || R < 32 || (R % 32) || rows_per_wg < 32 || (rows_per_wg % 32)) return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const Sched sc = default_sched();
    supdate_split_kernel<<<dim3(nchunks, npairs, batch), 256, 0, st>>>(sc, X, panel_stride, batch_stride, ns, D, R, rows_per_wg, Qfin, subact, done, nupd);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

// Test hook: the Gram matrix of the reduction alone.  mode 0: gram64_kernel; 1: the int8 digit path (scratch / ex as launch_gram_i8, rows in
// segments of at most seg_rows when >= 64).
int asvd_test_gram(const float* Xp, int64_t panel_stride, int64_t batch_stride, int nb, int m_pad, int batch, int mode, int seg_rows, double* G,
                   void* scratch, size_t scratch_bytes, int* ex, void* stream) {
    if (!Xp || !G || nb < 1 || m_pad < 32 || (m_pad % 32) || batch < 1) return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int n_pad = nb * PB;
    if (mode == 0) {
        gram64_kernel<<<dim3(nb, (unsigned)ceil_div64(nb, 4), batch), 256, 0, st>>>(Xp, panel_stride, batch_stride, nb, m_pad, G, n_pad, (int64_t)n_pad * n_pad);
        ASVD_HIP_CHECK(hipGetLastError());
        return ASVD_OK;
    }
    if (!scratch || !ex) return ASVD_E_BADARG;
    colmaxexp_kernel<<<dim3(nb, batch), 256, 0, st>>>(Xp, panel_stride, batch_stride, m_pad, n_pad, ex, nullptr);
    return launch_gram_i8(Xp, panel_stride, batch_stride, nb, m_pad, n_pad, batch, G, n_pad, (int64_t)n_pad * n_pad, (signed char*)scratch, scratch_bytes, ex,
                          nullptr, nullptr, nullptr, seg_rows, st);
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
    const int nsteps = grp ? grouped_steps(ns, group_bits_for(ns)) : pw2 - 1;
    *nsteps_out = nsteps;
    *npairs_out = npairs;
    if ((int64_t)nsteps * npairs > out_capacity || npairs > 1024) return ASVD_E_WORKSPACE;
    Sched sc = default_sched();
    sc.super_order = grp ? 2 : 1;
    if (grp) set_group_table(sc, ns, group_bits_for(ns));
    super_schedule_kernel<<<nsteps, npairs>>>(sc, ns, nsteps, npairs, out_dev);
    ASVD_HIP_CHECK(hipGetLastError());
    ASVD_HIP_CHECK(hipDeviceSynchronize());
    return ASVD_OK;
}

// Test hook: ONE launch of the fused update + next-step Gram kernel (supgram_kernel) on caller-built panels.  Din: [batch][npairs][128] squared column
// norms of every super-pair of step D in Q order (what the eigen-solve launch leaves in EvdV3::Din).  Gx: [batch][npairs][nchunks][6][1024]
// partial tiles of super-step E, indexed by E's pair slots.  E != D, both in [1, 2 * npairs).
int asvd_test_supgram(float* X, int64_t panel_stride, int64_t batch_stride, int ns, int D, int E, int R, int m_pad, int rows_per_wg,
                      const float* Qfin, const int* subact, const float* Din, float* Gx, const int* done, int* nupd, int nchunks, int npairs, int batch,
                      void* stream) {
    if (!X || !Qfin || !subact || !Din || !Gx || !done || !nupd || ns < 3 || npairs < 2 || D < 1 || E < 1 || D == E || D >= 2 * npairs || E >= 2 * npairs ||
        R < 32 || (R % 32) || (m_pad % 32) || rows_per_wg < 32 || (rows_per_wg % 32))
        return ASVD_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Sched sc = default_sched();
#ifdef ASVD_SG_TIMING
    if (getenv("ASVD_SG_ABLATE")) sc.meas = atoi(getenv("ASVD_SG_ABLATE"));  // tools/bench_supgram.py --timing --ablate N (measurement build only)
#endif
    ASVD_HIP_CHECK(hipFuncSetAttribute((const void*)supgram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SUPGRAM_SMEM_FLOATS * sizeof(float))));
    supgram_kernel<<<dim3(nchunks, (2 * npairs) / 4, batch), 512, SUPGRAM_SMEM_FLOATS * sizeof(float), st>>>(sc, X, panel_stride, batch_stride, ns, D, E, R, m_pad,
                                                                                                           rows_per_wg, Qfin, subact, Din, Gx, done, nupd, npairs);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

#ifdef ASVD_SG_TIMING
int asvd_test_sg_timing(unsigned long long* out_host) {  // [2][10] per-stage shader cycles (index 9: tiles) of the last supgram launch (timing builds only)
    ASVD_HIP_CHECK(hipDeviceSynchronize());
    ASVD_HIP_CHECK(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_sg_ts), 2 * 10 * sizeof(unsigned long long)));
    return ASVD_OK;
}
#endif

// Test hook (tests/test_gpu_evd_wave.py): the wave-local 64x64 eigen-solver alone.  G: [batch][64][64] symmetric (device); outputs
// Q [batch][64][64] (unsorted, unscaled), diag / rnk / cs [batch][64], Gout [batch][64][64] (the image the sweeps leave), meas [batch][2].
int asvd_test_evd_wave(const float* G, int batch, int sweeps, float* Q, float* diag, int* rnk, float* cs, float* Gout, float* meas, void* stream) {
    if (!G || !Q || !diag || !rnk || !cs || !Gout || !meas || batch < 1) return ASVD_E_BADARG;   // sweeps < 0: |sweeps| cross-only visits on the ring
    launch_evdw_test(batch, (hipStream_t)stream, G, sweeps, Q, diag, rnk, cs, Gout, meas);
    ASVD_HIP_CHECK(hipGetLastError());
    return ASVD_OK;
}

static int svd_batched_one(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                           const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                           float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                           int* info_host, void* stream) {
    Plan p;
    int rc = make_plan(batch, m, n, 0, 0, p);
    if (rc) return rc;
    if (g_prof_enabled) prof_begin();
    rc = -100;
    g_last_path = 0;
    if (tall_wanted(p)) {
        rc = svd_tall(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes,
                      info_host, stream);
        g_last_path |= (rc == -100) ? ASVD_PATH_REDUCE_FALLBACK : ASVD_PATH_REDUCED;
    }
    if (rc == -100)
        rc = svd_direct(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes,
                        info_host, stream, false);
    if (g_prof_enabled) prof_end();
    return rc;
}

static int svd_batched_entry(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                             const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                             float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                             int* info_host, void* stream) {
    if (!a_host || !S_host || !work || !dtype_ok(a_dtype) || lda < n) return ASVD_E_BADARG;
    if (cs_host && !dtype_ok(cs_dtype)) return ASVD_E_BADARG;
    if (batch < 1 || m < 1 || n < 1) return ASVD_E_BADARG;
    if (k < 1 || k > std::min(m, n)) return ASVD_E_BADARG;
    for (int b = 0; b < batch; ++b)
        if (!a_host[b] || !S_host[b]) return ASVD_E_BADARG;
    {
        int dev0 = 0;
        if (hipGetDevice(&dev0) == hipSuccess) presence_register(dev0);
    }
    // two halves on disjoint halves of the chip (see SplitCtx above)
    g_prof_last_split = false;
    if (split_applies(batch, m, n) && (!g_prof_enabled || g_prof_mode == 2) && g_call_cus == 0) {
        SplitCtx* ss = split_ctx();
        const int want_vectors = (U_host || V_host) ? 1 : 0;
        const int nb0 = (batch + 1) / 2, nb1 = batch / 2;
        size_t h0 = 0, h1 = 0;
        int rcw = ASVD_OK;
        if (ss) {
            g_call_cus = ss->cus;
            rcw = worksize_one(nb0, m, n, want_vectors, &h0);
            if (!rcw) rcw = worksize_one(nb1, m, n, want_vectors, &h1);
            g_call_cus = 0;
        }
        const size_t off1 = (h0 + 255) & ~(size_t)255;
        bool refused = !ss;
        if (ss && !rcw && work_bytes >= off1 + h1) {
            // not when another process computes on this device, not on a caller's CU-masked stream: such a call runs as ONE call on the caller's stream
            int dev = 0;
            ASVD_HIP_CHECK(hipGetDevice(&dev));
            if (split_device_shared(dev) || stream_is_cu_masked((hipStream_t)stream)) refused = true;
            else {
                const bool prof = g_prof_enabled;
                // everything the caller queued on its stream (weights, scale vectors) is visible to both halves
                hipEvent_t ev;
                ASVD_HIP_CHECK(hipEventCreateWithFlags(&ev, prof ? hipEventDefault : hipEventDisableTiming));
                ASVD_HIP_CHECK(hipEventRecord(ev, (hipStream_t)stream));
                ASVD_HIP_CHECK(hipStreamWaitEvent(ss->s[0], ev, 0));
                ASVD_HIP_CHECK(hipStreamWaitEvent(ss->s[1], ev, 0));
                const int cus = ss->cus;
                int rc1 = ASVD_E_HIP, path1 = 0;
                float ms1[NPROF] = {0};
                int ln1[NPROF] = {0};
                long long pairs1[3] = {0, 0, 0};
                std::vector<std::pair<float, float>> iv1;
                {
                    std::lock_guard<std::mutex> lk(ss->m);
                    ss->job = [&]() {
                        g_call_cus = cus;
                        g_prof_enabled = prof;
                        g_prof_base = prof ? ev : nullptr;
                        try {
                            rc1 = svd_batched_one(nb1, a_host + nb0, a_dtype, m, n, lda, cs_host ? cs_host + nb0 : nullptr, cs_dtype, U_host ? U_host + nb0 : nullptr,
                                                  S_host + nb0, V_host ? V_host + nb0 : nullptr, k, max_sweeps, tol, (char*)work + off1, h1,
                                                  info_host ? info_host + 4 * nb0 : nullptr, (void*)ss->s[1]);
                        } catch (...) { rc1 = ASVD_E_HIP; }
                        path1 = g_last_path;
                        if (prof) {
                            for (int i = 0; i < NPROF; ++i) { ms1[i] = g_prof_ms[i]; ln1[i] = g_prof_launches[i]; }
                            for (int i = 0; i < 3; ++i) pairs1[i] = g_prof_pairs[i];
                            iv1.swap(g_prof_iv8);
                        }
                        g_call_cus = 0;
                        g_prof_enabled = false;
                        g_prof_base = nullptr;
                    };
                    ss->has_job = true;
                    ss->job_done = false;
                }
                ss->cv.notify_all();
                int rc0 = ASVD_E_HIP;
                g_call_cus = cus;
                g_prof_base = prof ? ev : nullptr;
                try {
                    rc0 = svd_batched_one(nb0, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, h0, info_host,
                                          (void*)ss->s[0]);
                } catch (...) { rc0 = ASVD_E_HIP; }
                g_call_cus = 0;
                g_prof_base = nullptr;
                {
                    std::unique_lock<std::mutex> lk(ss->m);
                    ss->cv.wait(lk, [&] { return ss->job_done; });
                }
                g_last_path |= path1 | ASVD_PATH_SPLIT;
                (void)hipStreamSynchronize(ss->s[0]);
                (void)hipStreamSynchronize(ss->s[1]);
                if (prof) {
                    g_prof_last_split = true;
                    for (int i = 0; i < NPROF; ++i) {
                        g_prof_half_ms[0][i] = g_prof_ms[i]; g_prof_half_launches[0][i] = g_prof_launches[i];
                        g_prof_half_ms[1][i] = ms1[i]; g_prof_half_launches[1][i] = ln1[i];
                        g_prof_ms[i] += ms1[i]; g_prof_launches[i] += ln1[i];
                    }
                    for (int i = 0; i < 3; ++i) g_prof_pairs[i] += pairs1[i];
                    // class 8 of both halves on one time axis: summed durations, union, and the time both halves were inside such a launch
                    std::vector<std::pair<float, int>> edges;
                    float sum0 = 0, sum1 = 0;
                    for (auto& iv : g_prof_iv8) { edges.emplace_back(iv.first, +1); edges.emplace_back(iv.second, -1); sum0 += iv.second - iv.first; }
                    for (auto& iv : iv1) { edges.emplace_back(iv.first, +1); edges.emplace_back(iv.second, -1); sum1 += iv.second - iv.first; }
                    std::sort(edges.begin(), edges.end());
                    float uni = 0, both = 0, last = 0;
                    int depth = 0;
                    for (auto& e : edges) {
                        if (depth >= 1) uni += e.first - last;
                        if (depth >= 2) both += e.first - last;
                        last = e.first;
                        depth += e.second;
                    }
                    g_prof_overlap[0] = sum0; g_prof_overlap[1] = sum1; g_prof_overlap[2] = uni; g_prof_overlap[3] = both;
                }
                (void)hipEventDestroy(ev);
                if (rc0 < 0) return rc0;
                if (rc1 < 0) return rc1;
                return std::max(rc0, rc1);
            }
        }
        if (refused) {
            const int rcu = svd_batched_one(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes,
                                            info_host, stream);
            g_last_path |= ASVD_PATH_SPLIT_REFUSED;
            return rcu;
        }
    }
    return svd_batched_one(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes, info_host,
                           stream);
}

int asvd_svd_batched(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                     const void* const* cs_host, int cs_dtype, float* const* U_host, float* const* S_host,
                     float* const* V_host, int64_t k, int max_sweeps, float tol, void* work, size_t work_bytes,
                     int* info_host, void* stream) {
    try {   // the host driver allocates (std::vector): nothing may be thrown across the C boundary
        return svd_batched_entry(batch, a_host, a_dtype, m, n, lda, cs_host, cs_dtype, U_host, S_host, V_host, k, max_sweeps, tol, work, work_bytes, info_host,
                                 stream);
    } catch (...) {
        return ASVD_E_HIP;
    }
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
