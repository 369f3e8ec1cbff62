/*
 * asvd_hip.h — C ABI of libasvd_hip.so: the MI355X (gfx950) kernels behind the activation-aware SVD
 * compression path of ASVD4LLM.
 *
 * The reference (hahnyuan/ASVD4LLM) is pure Python and has no FFI layer; its arithmetic for this path
 * lives in eager torch ops.  Every entry point below replaces one group of those torch ops; the
 * reference lines are cited per function (paths relative to the reference tree).  A maintainer of the
 * reference binds these with ctypes from inside the same Python functions.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All data pointers are DEVICE pointers unless the name
 *     ends in `_host`.  The caller owns every buffer; the library never frees or retains them.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are enqueued on it.
 *     Functions documented "host-synchronous" call hipStreamSynchronize(stream) internally because
 *     they return host scalars (sweep counts / NaN flags) or steer iteration on device results.
 *   - return value: 0 = OK; negative = bad argument / runtime error (ASVD_E_*); positive = numerical
 *     condition (ASVD_N_*).  asvd_status_string() gives text.
 *   - matrices are row-major; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef ASVD_HIP_H
#define ASVD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types */
#define ASVD_F32 0
#define ASVD_F16 1
#define ASVD_BF16 2

/* status codes */
#define ASVD_OK 0
#define ASVD_E_BADARG (-1)
#define ASVD_E_WORKSPACE (-2)   /* workspace too small */
#define ASVD_E_HIP (-3)         /* a HIP runtime call failed */
#define ASVD_E_NODEVICE (-4)    /* no gfx950 device visible */
#define ASVD_N_NOCONV 1         /* Jacobi did not reach tol within max_sweeps (results still usable) */
#define ASVD_N_NAN 2            /* NaN/Inf seen in outputs */

/* sigma_fuse modes of SVDLinear.__init__ (modules/svd_linear.py:16-24) */
#define ASVD_FUSE_UV 0
#define ASVD_FUSE_U 1
#define ASVD_FUSE_V 2

/* activation-statistic modes of the calibration hook (act_aware_utils.py:64-74) */
#define ASVD_STAT_ABS_MEAN 0
#define ASVD_STAT_ABS_MAX 1
/* column mean of x^2 with x^2 rounded to the dtype of x first: `weight.grad.pow(2).mean(0)` of calib_fisher_info (act_aware_utils.py:30) */
#define ASVD_STAT_SQ_MEAN 2

int asvd_version(void);
const char* asvd_status_string(int status);
/* number of visible HIP devices with arch gfx950; <=0 means the library cannot run */
int asvd_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * K1/K2  calibration hook accumulator.
 * Replaces act_aware_utils.py:66-67 (abs_mean: `input[0].abs().mean(dim=-2).view(-1)`; `acc += ...`)
 * and :69-74 (abs_max: `.abs().amax(dim=-2)`; `torch.where(abs_max > acc, abs_max, acc)`).
 *   x   [rows, cols] row-major with leading dimension ld (the [1,T,C] / [T,C] hook input flattened)
 *   acc [cols] in acc_dtype (the reference keeps the activation dtype); caller zero-initialises it,
 *       which reproduces the reference's python-int 0 start exactly (0 + x == x; max(x,0) == x).
 * abs_mean: column sums accumulate in fp32, are divided by rows, rounded to acc_dtype, then added to
 *           acc in acc_dtype arithmetic (one rounding), i.e. the reference's two-op sequence.
 * abs_max : NaN in x never replaces acc (torch.where(nan > acc) is false).
 * sq_mean : as abs_mean with x*x (rounded to x's dtype, as torch's .pow(2) does) in place of |x|; x = weight.grad [out, in].
 * work: asvd_absstat_worksize bytes (fp32 partial sums). Asynchronous on `stream`.
 */
int asvd_absstat_worksize(int64_t rows, int64_t cols, size_t* bytes);
int asvd_absstat_accum(const void* x, int x_dtype, int64_t rows, int64_t cols, int64_t ld,
                       void* acc, int acc_dtype, int mode, void* work, size_t work_bytes, void* stream);
/* The two halves of asvd_absstat_accum, for Linears that are called with the SAME input tensor (q/k/v_proj, gate/up_proj: the
 * reference's hook re-reads X for each of them, act_aware_utils.py:78-81): asvd_absstat_partial makes the one pass over X
 * (fp32 partial column statistics into `work`), asvd_absstat_finalize applies them to one accumulator with the arithmetic
 * described above — call it once per Linear sharing the input.  Both asynchronous on `stream`. */
int asvd_absstat_partial(const void* x, int x_dtype, int64_t rows, int64_t cols, int64_t ld, int mode,
                         void* work, size_t work_bytes, void* stream);
int asvd_absstat_finalize(const void* work, size_t work_bytes, int64_t rows, int64_t cols,
                          void* acc, int acc_dtype, int mode, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3a  scale vector.  Replaces svd_linear.py:48-59:
 *   s = 1 * scaling_diag_matrix**alpha [* fisher_info**alpha]; s += 1e-6
 * evaluated in the dtype of the statistics (fp16 for an fp16 model: pow, product and +eps each round
 * to that dtype, as torch does).  fisher may be NULL.  out has the same dtype.  Asynchronous.
 */
int asvd_make_scale(const void* scaling, const void* fisher, int dtype, int64_t n, float alpha, float eps,
                    void* out, void* stream);
/* batched form: host arrays [batch] of device pointers (fisher_host may be NULL or hold NULL entries); one call for the
 * q/k/v/o (gate/up) Linears of a layer or a whole same-shape batch of asvd_svd_batched.  Asynchronous. */
int asvd_make_scale_batched(int batch, const void* const* scaling_host, const void* const* fisher_host, int dtype, int64_t n,
                            float alpha, float eps, void* const* out_host, void* stream);

/* K3b  w_scaled[i][j] = float(w[i][j]) * float(s[j]).  Replaces svd_linear.py:47,60
 * (`w = linear.weight.data.float(); w = w * scaling_diag_matrix.view(1,-1)`).  s may be NULL (plain
 * upcast).  Stand-alone form used by parity tests; asvd_svd fuses the same arithmetic into its pack
 * kernel.  Asynchronous. */
int asvd_scale_cols(const void* w, int w_dtype, int64_t m, int64_t n, int64_t ldw,
                    const void* s, int s_dtype, float* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  economy SVD in fp32:   A * diag(col_scale) = U diag(S) V^T .
 * Replaces the factorisation call at svd_linear.py:65 (`torch.svd_lowrank(w, q=rank)`; the parity
 * oracle named by BASELINE.json is the exact `torch.linalg.svd(w, full_matrices=False)`), and, with
 * U = V = NULL, the values-only `torch.svd(w.float(), compute_uv=False)` at sensitivity.py:101.
 *
 * Algorithm: the oriented matrix (rows >= cols) is reduced to a square one by a Cholesky-QR in fp64 (Gram matrix
 * by fp64 MFMA, columns ordered by norm), then one-sided block Jacobi runs on R^T: XOR pair schedule over 32-column panels
 * (panel counts that are not a power of two: a grouped schedule — XOR inside groups of 2..16 super-panels, round-robin over
 * the groups); dense sweeps work on 64-column super-panels, TWO launches per super-step: the wave-local 64x64 eigen-solves
 * (one wave per solve, the matrix in registers, both inner steps of a super-pair; from 2048 columns on the second inner step visits
 * only the cross pairs of its two panels, on a ring of interleaved positions: half the phases) and one 128-wide update pass fused with the
 * Gram tiles of the next step, in split-fp16 arithmetic (three products per fp32 product, power-of-two column scales from the
 * carried column norms; a problem that turns NaN on that path is repeated with the separate fp32 Gram / split-bf16 update
 * passes); tail sweeps rotate only the pairs a blocked X^T X snapshot marks.  Right vectors are the rotated columns, left
 * vectors X V by one GEMM, sigma_j = |X v_j| in fp64.  No vector is accumulated during the sweeps.  Problems too small or
 * rank-deficient for the reduction take the same sweeps on the matrix itself.
 *
 *   batch       number of same-shape problems solved concurrently (fills the 256 CUs)
 *   a_host      host array [batch] of device pointers to A_b  [m, n] row-major, leading dim lda
 *   cs_host     host array [batch] of device pointers to column scales s_b [n] (or NULL / NULL entries)
 *   U_host      host array [batch] of device pointers to U_b [m, k] row-major (NULL: no vectors)
 *   S_host      host array [batch] of device pointers to S_b [k] descending
 *   V_host      host array [batch] of device pointers to V_b [n, k] row-major (NULL: no vectors)
 *   k           number of leading singular triplets to write, 1 <= k <= min(m, n).  Convergence (tol) is enforced for these
 *               k leading triplets and their orthogonality against the rest; with k < min(m,n) the sweeps stop as soon as
 *               the leading part is done (the discarded tail converges last and would cost 2-4 more sweeps)
 *   max_sweeps  <=0: default (30);  tol <=0: default (1e-6) on max |cos(a_i, a_j)|
 *   info_host   optional host int[4*batch]: {status, sweeps, rotated pairs in last sweep, float bits of the last sweep's max |cos|}
 * Host-synchronous: the call returns when S/U/V are complete.  Sync points (csrc/svd_jacobi.hip): one after the reduction
 * (Cholesky breakdown flag), one per Jacobi sweep (convergence flags: four ints per problem), one more per sparse sweep
 * (the pair marks of the coupling snapshot go to the host, which packs them into rounds of disjoint pairs), one when a
 * problem finishes early (its `done` flag goes to the device), one at the end — 13 for a 4096 x 4096 call, 0.3 % of its time.
 * Everything is enqueued on `stream`; the one exception is the split of a large batch over two internal CU-masked streams
 * (asvd_svd_set_split below: what the library OWNS for it and when it refuses; the streams wait for everything queued on `stream` before the
 * call, and the call returns synchronised).  Concurrent calls from
 * different host threads on different streams and workspaces are safe: no shared mutable state — schedule tables and modes
 * travel by value in the kernel arguments, not in __constant__ memory; the profiling counters are per thread — and no kernel
 * uses scratch.  tests/test_gpu_concurrency.py holds the library to it (two threads x 8 x 4096^2 next to a stream of foreign
 * GEMMs, and calls with different pair schedules side by side: bit-identical results and sweep counts against serial runs).
 * Environment (read on the host, all optional; none changes results beyond rounding): ASVD_DEBUG (trace on stderr),
 * ASVD_DEBUG_HIST (pair-measure histogram per sweep), ASVD_ORDER=rr (round-robin instead of XOR pair order, single-level sweeps),
 * ASVD_TWOLEVEL=0, ASVD_SUPGRAM=0 (separate Gram / update passes), ASVD_SPARSE=0, ASVD_NO_REDUCE (skip the Cholesky-QR),
 * ASVD_EVDQ=0/1 (force the throughput / latency form of the eigen-solver), ASVD_EVDW_TRACE (stage stamps of the solver),
 * ASVD_SPREAD_FROM=<sweep> (line-spread order of the XOR distances from that dense sweep on: a measurement knob),
 * ASVD_SPLIT=0 (never split a batch over the two halves of the chip), ASVD_RING=0/1/2 + ASVD_RING_FROM=<sweep> (cross-only ring visits of the
 * eigen-solves: off / both inner steps / inner step 1 only — the default for >= 2048 columns is 2), ASVD_GRAM_I8=0 (Gram matrix of the reduction
 * with the fp64 matrix instructions instead of the exact int8 digit products, csrc/gram_i8.h), ASVD_NN_I8=0 (long-side product X V with six bf16
 * products per fp32 product instead of the int8 fixed-point form, csrc/nn_gemm_i8.h), ASVD_SNAP_I8=0 (coupling snapshot of the sparse sweeps with
 * three fp16 products per fp32 product instead of int8 digit planes, csrc/snapshot_i8.h); ASVD_GI_ORDER / ASVD_NI_ORDER=0/1/2 (block order of the two
 * int8 GEMM kernels: measurement knobs, tools/bench_i8_gemm.py).
 * Returns worst status over the batch.
 */
int asvd_svd_worksize(int batch, int64_t m, int64_t n, int want_vectors, size_t* bytes);
int asvd_svd_batched(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                     const void* const* cs_host, int cs_dtype,
                     float* const* U_host, float* const* S_host, float* const* V_host, int64_t k,
                     int max_sweeps, float tol, void* work, size_t work_bytes, int* info_host, void* stream);
/* Which path the last asvd_svd_batched / asvd_svd call of THIS host thread took: an OR of the bits below over the problems of the batch (both
 * halves of a split batch).  The results meet the same contract on every path; the bits are there so that a caller — and the parity tests —
 * can tell a silent change of arithmetic from the default one. */
#define ASVD_PATH_REDUCED 1          /* Cholesky-QR reduction + Jacobi on R^T (the default for >= 128 columns) ran to completion */
#define ASVD_PATH_REDUCE_FALLBACK 2  /* the Cholesky broke down (pivot <= 1e-13: numerically rank-deficient) -> the whole call ran the direct path */
#define ASVD_PATH_PLAIN_RETRY 4      /* a problem turned NaN in the split-fp16 fused kernel -> the sweeps were repeated with the separate passes */
#define ASVD_PATH_SPLIT 8            /* the batch ran as two halves on two CU-masked streams */
#define ASVD_PATH_SPLIT_REFUSED 16   /* the batch qualified for the split but ran as one call on the caller's stream: another process computes on the
                                        device or the caller's stream carries a CU mask of its own */
#define ASVD_PATH_GRAM_RETRY 32      /* the Cholesky broke down on the int8 Gram matrix (columns rounded to 2^-25 of their largest entry) and the reduction was
                                      * repeated with the fp64 Gram matrix of the exact columns (which either completes: REDUCED, or breaks down too: REDUCE_FALLBACK) */
int asvd_svd_get_last_path(void);
/* single-problem convenience wrapper (batch = 1) */
int asvd_svd(const void* a, int a_dtype, int64_t m, int64_t n, int64_t lda, const void* col_scale, int cs_dtype,
             float* U, float* S, float* V, int64_t k, int max_sweeps, float tol,
             void* work, size_t work_bytes, int* info_host, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4s  largest singular value only.  Replaces sensitivity.py:101-102 of calib_sensitivity_stable_rank
 * (`_, singular_values, _ = torch.svd(w.float(), compute_uv=False); spectral_norm = torch.max(singular_values)`):
 * the reference factorises the whole matrix and keeps one number.  Lanczos on W^T W (two matrix-vector passes
 * over W per step, fp32 vectors, fp64 recurrence coefficients); every 16 steps the largest Ritz value is found by
 * multisection on the Sturm count and the iteration stops when it moved by less than tol (relative, in sigma^2)
 * over the last 16 steps.  HBM/L2-bound: 2*m*n*sizeof(elem) bytes per step.
 *   a_host      host array [batch] of device pointers to W_b [m, n] row-major (leading dim lda), n <= 16384
 *   sigma_host  host array [batch] of device pointers to one float each (sigma_max of W_b)
 *   max_steps   <=0: default 1024 (rounded up to a multiple of 16);  tol <=0: default 1e-8
 *   info_host   optional host int[2*batch]: {status, Lanczos steps}
 * Host-synchronous (one stream sync per 16 steps).  Returns worst status over the batch (ASVD_N_NOCONV: value is a
 * lower bound that was still moving — callers fall back to asvd_svd_batched with k = 1).
 */
int asvd_sigma_max_worksize(int batch, int64_t m, int64_t n, int max_steps, size_t* bytes);
int asvd_sigma_max_batched(int batch, const void* const* a_host, int a_dtype, int64_t m, int64_t n, int64_t lda,
                           float* const* sigma_host, int max_steps, float tol, void* work, size_t work_bytes,
                           int* info_host, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5/K6  truncate to rank r, un-scale V, fuse sigma, transpose V, down-cast, NaN flag.
 * Replaces svd_linear.py:69-70 (`V = V / s.view(-1,1)`), :81-98 (NaN checks) and SVDLinear.__init__
 * :16-24 + the `.to(dtype)` at :102:
 *   UV: A = U[:, :r] * sqrt(S[:r])           B = (V[:, :r] / s[:,None]).T * sqrt(S[:r])[:,None]
 *   U : A = U[:, :r] * S[:r]                 B = (V[:, :r] / s[:,None]).T
 *   V : A = U[:, :r]                         B = (V[:, :r] / s[:,None]).T * S[:r][:,None]
 * all in fp32, then rounded once (RNE) to out_dtype.
 *   U [m, ldu>=r], S [>=r], V [n, ldv>=r] fp32;  s [n] in s_dtype or NULL;
 *   A [m, r] (ALinear.weight), B [r, n] (BLinear.weight) contiguous in out_dtype;
 *   nan_flags: device int[3] OR-ed with 1 where S / U / V hold a NaN within the first r columns
 *              (caller zero-initialises; NULL to skip).  Asynchronous.
 */
int asvd_truncate_split(const float* U, int64_t ldu, const float* S, const float* V, int64_t ldv,
                        const void* s, int s_dtype, int64_t m, int64_t n, int64_t r, int sigma_fuse,
                        void* A, void* B, int out_dtype, int* nan_flags, void* stream);
/* batched form over same-shape problems: host arrays [batch] of device pointers; nan_flags: device int[3 * batch] (3 per
 * problem, caller zero-initialises) or NULL.  Asynchronous. */
int asvd_truncate_split_batched(int batch, const float* const* U_host, int64_t ldu, const float* const* S_host,
                                const float* const* V_host, int64_t ldv, const void* const* s_host, int s_dtype, int64_t m,
                                int64_t n, int64_t r, int sigma_fuse, void* const* A_host, void* const* B_host, int out_dtype,
                                int* nan_flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K8  squared Frobenius norm (sensitivity.py:100 `torch.norm(w, p="fro") ** 2`), fp32 accumulate in
 * fixed order, result written to *out (device float).  work: asvd_fro_worksize bytes.  Asynchronous. */
int asvd_fro_worksize(int64_t m, int64_t n, size_t* bytes);
int asvd_fro_norm_sq(const void* w, int w_dtype, int64_t m, int64_t n, int64_t ldw, float* out,
                     void* work, size_t work_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K9  parity evidence (new; no reference counterpart — BASELINE.json north_star "reconstructed W <= 1e-3 Frobenius"):
 *        err2 = |W - A*B|_F^2 , w2 = |W|_F^2      (A, B = ALinear.weight, BLinear.weight of an SVDLinear: svd_linear.py:8-24)
 *   W [m, n] in w_dtype (ldw), A [m, r], B [r, n] contiguous in ab_dtype.  F16 / BF16 factors: tiled GEMM on the fp16 / bf16 matrix pipe
 *   (products of 16-bit factors are exact there, fp32 accumulation) fused with the squared-difference reduction in fp64, 2 m n r flop.  When
 *   n % 8 == 0 and B is 16-byte aligned (every Linear of the models this path serves) the GEMM reads the factors AS STORED — A rows at any 2-byte
 *   alignment (odd ranks), B transposed on its way into LDS: ONE GEMM launch + the ordered sum of its partials; A is read in aligned 4-byte words, so
 *   its allocation must be readable up to the next 4-byte boundary.  Otherwise: zero-padded K-contiguous copies in the workspace first (two more
 *   launches).  F32 factors: fp32 MFMA, one wave per 32x32 tile.  out: device double[2] = {err2, w2}.  work: asvd_reconstruct_worksize(m, n, r)
 *   bytes (partial sums + room for the padded copies).  Asynchronous. */
int asvd_reconstruct_worksize(int64_t m, int64_t n, int64_t r, size_t* bytes);
int asvd_reconstruct_err(const void* W, int w_dtype, int64_t ldw, const void* A, const void* B, int ab_dtype,
                         int64_t m, int64_t n, int64_t r, double* out, void* work, size_t work_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K10  fused SVDLinear forward for few tokens (modules/svd_linear.py:105-109 `y = self.BLinear(inp); y = self.ALinear(y)`):
 *        y[T, N] = fp16( fp16(x Bp^T) Ap^T + bias )
 *   ONE persistent launch (phase 1 z = x Bp^T, in-kernel grid barrier, phase 2 y = z Ap^T + bias); the r-wide intermediate z is
 *   rounded to fp16 exactly where BLinear's output is, lives in `work` (agent-scope stores / loads, no cache-wide fence) and is
 *   consumed inside the launch; B and A cross HBM once.  T <= 4: a pair of fused GEMVs (v_dot2_f32_f16); larger T: fp16 MFMA tiles.
 *   x [T, K] fp16 contiguous, 1 <= T <= ASVD_LOWRANK_MAX_TOKENS;  K % 64 == 0;
 *   Bp [rp, K] = BLinear.weight [r, K] with zero rows appended, Ap [N, rp] = ALinear.weight [N, r] with zero columns appended,
 *   rp = asvd_lowrank_padded_rank(r) (multiple of 64);  bias [N] fp16 or NULL;  y [T, N] fp16 contiguous.
 *   work: asvd_lowrank_work_bytes(T, rp) bytes, 16-byte aligned, whose first 256 bytes the caller zeroes ONCE (barrier state; the
 *   kernel leaves them reusable); one `work` must not be used by two launches in flight at the same time.  Asynchronous.
 *   For larger T the reference's two GEMMs are the right shape (weights amortised over tokens); this entry refuses them. */
#define ASVD_LOWRANK_MAX_TOKENS 256
int64_t asvd_lowrank_padded_rank(int64_t r);
size_t asvd_lowrank_work_bytes(int64_t T, int64_t rp);
int asvd_lowrank_forward_f16(const void* x, int64_t T, const void* Bp, const void* Ap, const void* bias, int64_t N, int64_t K,
                             int64_t rp, void* y, void* work, size_t work_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * C1  collective for the one exchange step of the multi-GPU path: all-gather of the per-layer sensitivities over
 * RCCL (xGMI).  The reference has no distributed path; a maintainer who binds only this library gets the collective here, the
 * Python host side uses torch.distributed (same RCCL) for it.  RCCL is resolved at run time from the process (torch's
 * librccl when loaded, else librccl.so) — the library has no link-time dependency on it.
 *   asvd_comm_init     rank 0 creates the unique id and writes it to `id_path` (a file visible to all ranks of the node); the
 *                      other ranks wait for the file (up to timeout_s) and read it; all call ncclCommInitRank on `device`.
 *   asvd_comm_allgather_f32 / _f64   send [count] values, recv [count * nranks], device pointers, asynchronous on `stream`.
 *   asvd_comm_destroy  releases the communicator (rank 0 removes the id file).
 * Return ASVD_E_HIP when RCCL is missing or a call fails. */
int asvd_comm_init(void** comm_out, int rank, int nranks, int device, const char* id_path, int timeout_s);
int asvd_comm_allgather_f32(void* comm, const float* send, float* recv, int64_t count, void* stream);
int asvd_comm_allgather_f64(void* comm, const double* send, double* recv, int64_t count, void* stream);
int asvd_comm_destroy(void* comm);

/* ---------------------------------------------------------------------------------------------
 * Instrumentation: per-kernel-class wall time of the last asvd_svd_batched call, measured with HIP
 * events on the call's stream when enabled.  classes: 0 pack + Cholesky-QR reduction, 1 two-level Gram pass (sgram6),
 * 2 eigen-solves, 3 two-level update pass (supdate), 4 finalize, 5 snapshot (the blocked X^T X pass that opens a sparse sweep),
 * 6 single-level Gram, 7 single-level update, 8 fused two-level update + next-step Gram pass (supgram).
 * ms_host: float[9] total milliseconds; launches_host: int[9].  */
/* enabled: 0 off; 1 on, and a profiled call is never split (every kernel alone on the whole chip: what a rocprofv3 pass with ASVD_SPLIT=0 sees);
 * 2 on, and the call keeps the split of asvd_svd_set_split — both halves time their own launches on their own stream, asvd_svd_get_profile returns
 * the sums over both halves, asvd_svd_get_split_profile the halves one by one. */
void asvd_svd_set_profiling(int enabled);
int asvd_svd_get_profile(float* ms_host, int* launches_host);
/* The last call profiled in mode 2, by half: ms_host float[2*9], launches_host int[2*9] (half h, class c at [h*9 + c]); overlap_host float[4] =
 * {summed milliseconds inside the fused update + Gram launches (class 8) of half 0, of half 1, the UNION of those intervals on one time axis, the
 * time BOTH halves were inside such a launch}.  Returns 1 when that call ran split, 0 when it did not (the arrays are then stale). */
int asvd_svd_get_split_profile(float* ms_host, int* launches_host, float* overlap_host);
/* The split of a batch over the two halves of the chip, and what the library owns for it.
 * asvd_svd_batched runs a batch of >= 4 problems with >= 3072 columns as two halves, each on an internal stream masked to one half of the CUs
 * (hipExtStreamCreateWithCUMask): the eigen-solve launches of one half (VALU-bound) then overlap the HBM-bound update launches of the other,
 * which two launches of one stream never do.
 * OWNERSHIP — asvd_svd_batched (and asvd_svd, which calls it) is the only entry point of this header that creates anything behind the caller's
 * back.  Per CALLING HOST THREAD and device, at that
 * thread's first split call, the library creates (a) two CU-masked streams (hipExtStreamCreateWithCUMask has no flags argument: they are
 * ordinary blocking streams, i.e. they also order against the legacy NULL stream) and (b) ONE worker thread (it runs the second half; the
 * first half runs on the calling thread); both are released when the calling thread ends.  Per thread, so that two host threads making split
 * calls at once stay independent and the decision to split never depends on timing (results are those of the serial calls, bit for bit).
 * Per process and device, from the first asvd_svd_batched call of any size, (c) one file descriptor on /dev/shm/asvd_hip_split.<pci bus id>
 * holding a one-byte POSIX record lock: the presence of this process on the device.  No other call creates threads, streams or files;
 * workspaces and outputs are always the caller's.
 * REFUSALS — such a batch runs as ONE call on the caller's stream (ASVD_PATH_SPLIT_REFUSED) when another PROCESS that loaded this library
 * computes on the same device (the lock file: two ranks on one GPU would otherwise both claim "the first half + the second half") or when the
 * caller's stream is itself CU-masked (hipExtStreamGetCUMask).  A call profiled in mode 1 and a call under asvd_svd_set_call_cus are never split.
 *   asvd_svd_set_split(mode)   process-wide: 0 never split; 1 split when the rules above allow; -1 (default) as 1 unless ASVD_SPLIT=0 is set.
 *   asvd_svd_set_call_cus(cus) CUs the asvd_svd_batched calls of THIS host thread may use (0 = the whole device, the default): a caller that
 *                              drives calls on CU-masked streams of its own announces the count so that the launch geometry is sized for it. */
void asvd_svd_set_split(int mode);
void asvd_svd_set_call_cus(int cus);
/* Test hook: one launch of the two-level update kernel ([X_S X_T] <- [X_S X_T] Qfin for every super-pair of XOR step D) on
 * caller-built panels X [batch][nb][R][32]; Qfin [batch][npairs][128*128]; subact [batch][npairs][4] (pair updated when any flag is
 * set); done, nupd [batch] ints.  Used by tests/test_gpu_twolevel.py only. */
int asvd_test_supdate(float* X, int64_t panel_stride, int64_t batch_stride, int ns, int D, int R, int rows_per_wg,
                      const float* Qfin, const int* subact, const int* done, int* nupd, int nchunks, int npairs, int batch, void* stream);
/* Test hook: one launch of the fused kernel of the two-level sweep (update of XOR super-step D + partial Gram tiles of super-step E,
 * E != D) on caller-built panels.  Gx [batch][npairs][nchunks][6][32*32]: tiles [0,2] [0,3] [1,2] [1,3] [0,1] [2,3] of E's super-pairs
 * over the first m_pad rows, one partial per row chunk.  Din [batch][npairs][128]: squared norms of the 128 columns of every super-pair
 * of step D BEFORE the update, in the pair's column order (panels 2S, 2S+1, 2T, 2T+1; zeros for an absent member) — the power-of-two
 * column scales of the split-fp16 arithmetic come from them (in the library the eigen-solve launch of the step leaves them behind).
 * Other arguments as asvd_test_supdate.  tests/test_gpu_twolevel.py only. */
int asvd_test_supgram(float* X, int64_t panel_stride, int64_t batch_stride, int ns, int D, int E, int R, int m_pad, int rows_per_wg,
                      const float* Qfin, const int* subact, const float* Din, float* Gx, const int* done, int* nupd, int nchunks, int npairs,
                      int batch, void* stream);
/* Test hook: the Gram matrix G = X^T X of the Cholesky-QR reduction alone, on caller-built panels X [batch][nb][m_pad][32] (m_pad a multiple of
 * 32).  G [batch][32 nb][32 nb] doubles, upper 32-blocks written.  mode 0: fp64 matrix instructions (gram64_kernel); mode 1: the exact int8
 * digit path the library uses unless ASVD_GRAM_I8=0 (csrc/gram_i8.h): scratch >= 3 * 32 nb * 64 * batch bytes of device memory (digit planes:
 * the rows go in segments of what fits, at most seg_rows when seg_rows >= 64), ex = 32 nb ints per problem (column exponents, left behind).
 * tests/test_gpu_gram_i8.py only. */
int asvd_test_gram(const float* Xp, int64_t panel_stride, int64_t batch_stride, int nb, int m_pad, int batch, int mode, int seg_rows, double* G,
                   void* scratch, size_t scratch_bytes, int* ex, void* stream);
/* Test hook: the super-panel pair schedule of the two-level sweeps for `ns` super-panels (grouped != 0: the grouped order where it applies,
 * else XOR).  out_dev: device int[out_capacity] >= nsteps * npairs; out[step * npairs + k] = (S << 16) | T or -1 (empty slot).
 * tests/test_gpu_twolevel.py only. */
int asvd_test_super_schedule(int ns, int grouped, int* out_dev, int out_capacity, int* nsteps_out, int* npairs_out);
/* Test hook: the wave-local 64x64 symmetric eigen-solver of the sweeps (csrc/evd_wave.h) alone, one wave per matrix, `sweeps` full
 * inner sweeps.  G [batch][64][64] symmetric; Q [batch][64][64] accumulated rotations (column = position, unsorted, unscaled);
 * diag / rnk / cs [batch][64]: eigenvalue estimate, descending-order rank and 1/|q_c| of every position; Gout: the image after the
 * sweeps; meas [batch][2]: the two off-diagonal measures of the input.  tests/test_gpu_evd_wave.py only. */
int asvd_test_evd_wave(const float* G, int batch, int sweeps, float* Q, float* diag, int* rnk, float* cs, float* Gout, float* meas, void* stream);
/* counts_host: long long[3] = {32-column panel-pair visits (one 64x64 eigen-solve each), pairs actually rotated, 128-column
 * super-pairs updated by the two-level sweeps (one 128-wide update pass over the rows each)} summed over the sweeps and problems of
 * the last profiled call: the algorithmic byte counts of the streaming kernels follow from these. */
int asvd_svd_get_pair_counts(long long* counts_host);
/* host wall time (ms) and rotated pairs of every Jacobi sweep of the last profiled call, whole batch together (the call
 * synchronises once per sweep).  Fills at most `cap` entries; returns the number of sweeps. */
int asvd_svd_get_sweep_times(float* ms_host, long long* rotated_host, int cap);

#ifdef __cplusplus
}
#endif
#endif /* ASVD_HIP_H */
