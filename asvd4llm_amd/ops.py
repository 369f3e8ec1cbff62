"""torch-tensor front end of the C ABI (include/asvd_hip.h).  torch is used for device memory and streams only —
every arithmetic step below runs in libasvd_hip.so.  Tensors must live on a gfx950 device; nothing falls back to CPU."""
import ctypes
import os

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}")


def _dev(t, name):
    if not t.is_cuda:
        raise L.AsvdHipError(f"{name} must be a device tensor (got {t.device}); the ASVD hot path has no CPU fallback")
    return t


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def _on(device):
    """context that makes `device` current — nothing at all when it already is (torch.cuda.device costs several microseconds per call,
    which is most of a decode-sized launch)"""
    return _NOCTX if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _work(nbytes, device):
    w = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    fill = os.environ.get("ASVD_DEBUG_WORKFILL")  # debug: poison (255 -> NaN patterns) or zero the workspace to expose reads of unwritten memory
    if fill is not None:
        w.fill_(int(fill))
    return w


def absstat_accum(x2d, acc, method):
    """acc (in place) <- hook update with |x| column statistic over rows of x2d [rows, cols] (act_aware_utils.py:64-74)"""
    lib = L.load(True)
    _dev(x2d, "x"), _dev(acc, "acc")
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and acc.is_contiguous() and acc.numel() == x2d.shape[1]
    rows, cols = x2d.shape
    mode = L.STAT_SQ_MEAN if method == "sq_mean" else (L.STAT_ABS_MEAN if "abs_mean" in method else L.STAT_ABS_MAX)
    nb = ctypes.c_size_t()
    L.check(lib.asvd_absstat_worksize(rows, cols, ctypes.byref(nb)), "asvd_absstat_worksize")
    work = _work(nb.value, x2d.device)
    with _on(x2d.device):
        L.check(lib.asvd_absstat_accum(_ptr(x2d), _dt(x2d), rows, cols, x2d.stride(0), _ptr(acc), _dt(acc), mode, _ptr(work),
                                       work.numel(), _stream(x2d)), "asvd_absstat_accum")
    return acc


def _stat_mode(method):
    return L.STAT_SQ_MEAN if method == "sq_mean" else (L.STAT_ABS_MEAN if "abs_mean" in method else L.STAT_ABS_MAX)


def absstat_partial(x2d, method):
    """first half of absstat_accum: ONE pass over x2d, fp32 partial statistics in a workspace tensor (returned) that
    absstat_finalize applies to any number of accumulators (Linears that share this input)"""
    lib = L.load(True)
    _dev(x2d, "x")
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    rows, cols = x2d.shape
    nb = ctypes.c_size_t()
    L.check(lib.asvd_absstat_worksize(rows, cols, ctypes.byref(nb)), "asvd_absstat_worksize")
    work = _work(nb.value, x2d.device)
    with _on(x2d.device):
        L.check(lib.asvd_absstat_partial(_ptr(x2d), _dt(x2d), rows, cols, x2d.stride(0), _stat_mode(method), _ptr(work), work.numel(), _stream(x2d)),
                "asvd_absstat_partial")
    return work


def absstat_finalize(work, rows, cols, acc, method):
    lib = L.load(True)
    _dev(acc, "acc")
    assert acc.is_contiguous() and acc.numel() == cols
    with _on(acc.device):
        L.check(lib.asvd_absstat_finalize(_ptr(work), work.numel(), rows, cols, _ptr(acc), _dt(acc), _stat_mode(method), _stream(acc)),
                "asvd_absstat_finalize")
    return acc


def make_scale(scaling, fisher=None, alpha=1.0, eps=1e-6):
    """s = scaling**alpha [* fisher**alpha] + eps in the dtype of `scaling` (svd_linear.py:48-59)"""
    lib = L.load(True)
    _dev(scaling, "scaling")
    scaling = scaling.contiguous()
    if fisher is not None:
        fisher = fisher.to(scaling.dtype).contiguous()
    out = torch.empty_like(scaling)
    with torch.cuda.device(scaling.device):
        L.check(lib.asvd_make_scale(_ptr(scaling), _ptr(fisher), _dt(scaling), scaling.numel(), float(alpha), float(eps), _ptr(out),
                                    _stream(scaling)), "asvd_make_scale")
    return out


def make_scale_batched(scalings, fishers=None, alpha=1.0, eps=1e-6):
    """batched make_scale (asvd_make_scale_batched): one ABI call for a list of same-length, same-dtype statistics vectors"""
    lib = L.load(True)
    B = len(scalings)
    scalings = [_dev(s, "scaling").contiguous() for s in scalings]
    n, dt = scalings[0].numel(), scalings[0].dtype
    assert all(s.numel() == n and s.dtype == dt for s in scalings)
    if fishers is not None:
        fishers = [None if f is None else f.to(dt).contiguous() for f in fishers]
    outs = [torch.empty_like(s) for s in scalings]
    arr = ctypes.c_void_p * B
    f_p = arr(*[(f.data_ptr() if f is not None else None) for f in fishers]) if fishers is not None else None
    with torch.cuda.device(scalings[0].device):
        L.check(lib.asvd_make_scale_batched(B, arr(*[s.data_ptr() for s in scalings]), f_p, _dt(scalings[0]), n, float(alpha), float(eps),
                                            arr(*[o.data_ptr() for o in outs]), _stream(scalings[0])), "asvd_make_scale_batched")
    return outs


def scale_cols(w, s=None):
    """fp32 w * s[None, :] (svd_linear.py:47,60)"""
    lib = L.load(True)
    _dev(w, "w")
    assert w.dim() == 2 and w.stride(1) == 1
    m, n = w.shape
    out = torch.empty((m, n), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        L.check(lib.asvd_scale_cols(_ptr(w), _dt(w), m, n, w.stride(0), _ptr(s), _dt(s) if s is not None else 0, _ptr(out), n,
                                    _stream(w)), "asvd_scale_cols")
    return out


class SvdInfo:
    """per-problem {status, sweeps, pairs rotated in the last sweep} + the path bits of the CALL the problem was part of
    (asvd_svd_get_last_path: reduced / reduce_fallback / plain_retry / split / split_refused / gram_retry)"""

    def __init__(self, status, sweeps, last_rot, path=0):
        self.status, self.sweeps, self.last_rotated_pairs, self.path = status, sweeps, last_rot, path
        self.reduced = bool(path & L.PATH_REDUCED)
        self.reduce_fallback = bool(path & L.PATH_REDUCE_FALLBACK)
        self.plain_retry = bool(path & L.PATH_PLAIN_RETRY)
        self.split = bool(path & L.PATH_SPLIT)
        self.split_refused = bool(path & L.PATH_SPLIT_REFUSED)
        self.gram_retry = bool(path & L.PATH_GRAM_RETRY)

    def __repr__(self):
        return (f"SvdInfo(status={self.status}, sweeps={self.sweeps}, last_rotated_pairs={self.last_rotated_pairs}, reduced={self.reduced}, "
                f"reduce_fallback={self.reduce_fallback}, plain_retry={self.plain_retry}, split={self.split}, gram_retry={self.gram_retry})")


def svd_batched(mats, col_scales=None, k=None, want_vectors=True, max_sweeps=0, tol=0.0):
    """Economy SVD of a list of same-shape device matrices: mats[b] * diag(col_scales[b]) = U S V^T.
    Returns (U list [m,k], S list [k], V list [n,k], infos).  U/V lists are None when want_vectors is False."""
    lib = L.load(True)
    B = len(mats)
    assert B >= 1
    m, n = mats[0].shape
    dev = mats[0].device
    for a in mats:
        _dev(a, "matrix")
        assert a.shape == (m, n) and a.dtype == mats[0].dtype and a.stride(1) == 1 and a.stride(0) == mats[0].stride(0)
    kmax = min(m, n)
    k = kmax if k is None else int(k)
    assert 1 <= k <= kmax
    nb = ctypes.c_size_t()
    L.check(lib.asvd_svd_worksize(B, m, n, 1 if want_vectors else 0, ctypes.byref(nb)), "asvd_svd_worksize")
    work = _work(nb.value, dev)
    S = [torch.empty(k, dtype=torch.float32, device=dev) for _ in range(B)]
    U = [torch.empty((m, k), dtype=torch.float32, device=dev) for _ in range(B)] if want_vectors else None
    V = [torch.empty((n, k), dtype=torch.float32, device=dev) for _ in range(B)] if want_vectors else None
    arr = ctypes.c_void_p * B
    a_p = arr(*[a.data_ptr() for a in mats])
    s_p = arr(*[s.data_ptr() for s in S])
    u_p = arr(*[u.data_ptr() for u in U]) if want_vectors else None
    v_p = arr(*[v.data_ptr() for v in V]) if want_vectors else None
    if col_scales is not None:
        for s in col_scales:
            _dev(s, "col_scale")
            assert s.numel() == n and s.is_contiguous() and s.dtype == col_scales[0].dtype
        c_p = arr(*[s.data_ptr() for s in col_scales])
        cdt = _dt(col_scales[0])
    else:
        c_p, cdt = None, 0
    info = (ctypes.c_int * (4 * B))()
    with torch.cuda.device(dev):
        rc = lib.asvd_svd_batched(B, a_p, _dt(mats[0]), m, n, mats[0].stride(0), c_p, cdt, u_p, s_p, v_p, k, int(max_sweeps),
                                  float(tol), _ptr(work), work.numel(), info, _stream(mats[0]))
        path = int(lib.asvd_svd_get_last_path())
    L.check(rc, "asvd_svd_batched")
    infos = [SvdInfo(info[4 * b], info[4 * b + 1], info[4 * b + 2], path) for b in range(B)]
    return U, S, V, infos


def svd(a, col_scale=None, k=None, want_vectors=True, max_sweeps=0, tol=0.0):
    U, S, V, infos = svd_batched([a], None if col_scale is None else [col_scale], k, want_vectors, max_sweeps, tol)
    if want_vectors:
        return U[0], S[0], V[0], infos[0]
    return None, S[0], None, infos[0]


SIGMA_MAX_COLS = 16384  # asvd_sigma_max_batched keeps the Lanczos vector in LDS


def sigma_max_batched(mats, max_steps=0, tol=0.0):
    """Largest singular value of each matrix of a same-shape list (Lanczos, asvd_sigma_max_batched).
    Returns (list of 0-dim fp32 device tensors, list of (status, steps)).  Matrices with more than SIGMA_MAX_COLS columns,
    and any problem the Lanczos iteration reports as not converged, go through the values-only k=1 mode of the Jacobi SVD."""
    lib = L.load(True)
    B = len(mats)
    assert B >= 1
    m, n = mats[0].shape
    dev = mats[0].device
    for a in mats:
        _dev(a, "matrix")
        assert a.shape == (m, n) and a.dtype == mats[0].dtype and a.stride(1) == 1 and a.stride(0) == mats[0].stride(0)
    if n > SIGMA_MAX_COLS:
        _, S, _, infos = svd_batched(mats, None, k=1, want_vectors=False)
        return [s[0] for s in S], [(i.status, -i.sweeps) for i in infos]
    nb = ctypes.c_size_t()
    L.check(lib.asvd_sigma_max_worksize(B, m, n, int(max_steps), ctypes.byref(nb)), "asvd_sigma_max_worksize")
    work = _work(nb.value, dev)
    out = torch.empty(B, dtype=torch.float32, device=dev)
    arr = ctypes.c_void_p * B
    a_p = arr(*[a.data_ptr() for a in mats])
    o_p = arr(*[out.data_ptr() + 4 * b for b in range(B)])
    info = (ctypes.c_int * (2 * B))()
    with torch.cuda.device(dev):
        rc = lib.asvd_sigma_max_batched(B, a_p, _dt(mats[0]), m, n, mats[0].stride(0), o_p, int(max_steps), float(tol), _ptr(work),
                                        work.numel(), info, _stream(mats[0]))
    L.check(rc, "asvd_sigma_max_batched")
    sig = [out[b] for b in range(B)]
    infos = [(info[2 * b], info[2 * b + 1]) for b in range(B)]
    redo = [b for b in range(B) if infos[b][0] == 1]  # ASVD_N_NOCONV: still moving after max_steps
    if redo:
        _, S, _, jinfos = svd_batched([mats[b] for b in redo], None, k=1, want_vectors=False)
        for b, s, ji in zip(redo, S, jinfos):
            sig[b] = s[0]
            infos[b] = (ji.status, -ji.sweeps)
    return sig, infos


def truncate_split(U, S, V, s, r, sigma_fuse, out_dtype):
    """(A [m,r], B [r,n], nan_flags[3]) per SVDLinear.__init__ + un-scaling (svd_linear.py:69-70,16-24,102)"""
    lib = L.load(True)
    _dev(U, "U")
    m, n = U.shape[0], V.shape[0]
    assert U.stride(1) == 1 and V.stride(1) == 1 and U.shape[1] >= r and V.shape[1] >= r and S.numel() >= r
    A = torch.empty((m, r), dtype=out_dtype, device=U.device)
    Bm = torch.empty((r, n), dtype=out_dtype, device=U.device)
    flags = torch.zeros(3, dtype=torch.int32, device=U.device)
    with torch.cuda.device(U.device):
        L.check(lib.asvd_truncate_split(_ptr(U), U.stride(0), _ptr(S), _ptr(V), V.stride(0), _ptr(s), _dt(s) if s is not None else 0,
                                        m, n, r, L.FUSE[sigma_fuse], _ptr(A), _ptr(Bm), _DT[out_dtype], _ptr(flags), _stream(U)),
                "asvd_truncate_split")
    return A, Bm, flags


def truncate_split_batched(Us, Ss, Vs, ss, r, sigma_fuse, out_dtype):
    """batched truncate_split (asvd_truncate_split_batched) over same-shape problems: returns ([A], [B], flags [batch, 3])"""
    lib = L.load(True)
    B = len(Us)
    m, n = Us[0].shape[0], Vs[0].shape[0]
    dev = Us[0].device
    for U, S, V in zip(Us, Ss, Vs):
        _dev(U, "U")
        assert U.shape[0] == m and V.shape[0] == n and U.stride(1) == 1 and V.stride(1) == 1 and U.stride(0) == Us[0].stride(0) and V.stride(0) == Vs[0].stride(0)
        assert U.shape[1] >= r and V.shape[1] >= r and S.numel() >= r
    As = [torch.empty((m, r), dtype=out_dtype, device=dev) for _ in range(B)]
    Bs = [torch.empty((r, n), dtype=out_dtype, device=dev) for _ in range(B)]
    flags = torch.zeros((B, 3), dtype=torch.int32, device=dev)
    arr = ctypes.c_void_p * B
    s_p = arr(*[(x.data_ptr() if x is not None else None) for x in ss]) if ss is not None else None
    sdt = _dt(ss[0]) if ss is not None and ss[0] is not None else 0
    with torch.cuda.device(dev):
        L.check(lib.asvd_truncate_split_batched(B, arr(*[u.data_ptr() for u in Us]), Us[0].stride(0), arr(*[x.data_ptr() for x in Ss]),
                                                arr(*[v.data_ptr() for v in Vs]), Vs[0].stride(0), s_p, sdt, m, n, r, L.FUSE[sigma_fuse],
                                                arr(*[a.data_ptr() for a in As]), arr(*[b.data_ptr() for b in Bs]), _DT[out_dtype], _ptr(flags),
                                                _stream(Us[0])), "asvd_truncate_split_batched")
    return As, Bs, flags


LOWRANK_MAX_TOKENS = 256  # ASVD_LOWRANK_MAX_TOKENS


def lowrank_pack(A, B):
    """Pad ALinear.weight [N, r] / BLinear.weight [r, K] (fp16, CUDA) to the layout asvd_lowrank_forward_f16 streams:
    (Ap [N, rp], Bp [rp, K], work) with rp = asvd_lowrank_padded_rank(r); work carries the zeroed barrier words."""
    lib = L.load(True)
    _dev(A, "A")
    N, r = A.shape
    K = B.shape[1]
    assert A.dtype == torch.float16 and B.dtype == torch.float16 and B.shape[0] == r
    if K % 64:
        raise ValueError(f"fused low-rank forward needs in_features % 64 == 0 (got {K})")
    rp = int(lib.asvd_lowrank_padded_rank(r))
    Ap = torch.zeros((N, rp), dtype=torch.float16, device=A.device)
    Ap[:, :r] = A
    Bp = torch.zeros((rp, K), dtype=torch.float16, device=A.device)
    Bp[:r] = B
    work = torch.zeros(int(lib.asvd_lowrank_work_bytes(LOWRANK_MAX_TOKENS, rp)), dtype=torch.uint8, device=A.device)
    return Ap, Bp, work


_LOWRANK_CALLS = {}   # launches since the last give-up check, PER WORKSPACE (every SVDLinear owns one): data_ptr -> count


def lowrank_check(work):
    """read (one host sync) and clear the give-up word of a fused-forward workspace; raises when a launch since the last check left its grid barrier"""
    _LOWRANK_CALLS[work.data_ptr()] = 0
    flags = work[:16].view(torch.int32)
    if int(flags[2].item()) != 0:
        flags[2] = 0
        raise L.AsvdHipError("asvd_lowrank_forward_f16: a workgroup gave up at the in-kernel grid barrier (the launch was not fully resident); "
                             "its output was poisoned with NaN.  Use the two nn.Linear GEMMs (fused_forward = False) next to persistent kernels of other streams.")


def lowrank_forward(x2d, Ap, Bp, bias, work):
    """y = fp16(fp16(x Bp^T) Ap^T + bias) in one launch (K10, svd_linear.py:105-109) for x2d [T <= 256, K] fp16"""
    lib = L.load(True)
    _dev(x2d, "x")
    T, K = x2d.shape
    N, rp = Ap.shape
    assert x2d.dtype == torch.float16 and x2d.is_contiguous() and Bp.shape == (rp, K) and 1 <= T <= LOWRANK_MAX_TOKENS
    y = torch.empty((T, N), dtype=torch.float16, device=x2d.device)
    with _on(x2d.device):
        L.check(lib.asvd_lowrank_forward_f16(_ptr(x2d), T, _ptr(Bp), _ptr(Ap), _ptr(bias), N, K, rp, _ptr(y), _ptr(work), work.numel(),
                                             _stream(x2d)), "asvd_lowrank_forward_f16")
    # word 2 of the barrier state: some workgroup left the in-kernel grid barrier without its peers (the workgroups of the launch were not all
    # resident); the kernel has poisoned what it wrote of y with NaN in that case.  The word is STICKY (only the host clears it), so it does not
    # have to be read after every launch: reading it is a host sync of ~18 us on a 21 us call (round 4 measured the fused forward "slower than
    # two GEMMs" under ASVD_STRICT for exactly that reason).  ASVD_STRICT checks every 64th launch OF EACH WORKSPACE (the counter is keyed by
    # the workspace: with one global counter only the modules that happened to make a 64th call were ever checked — ADVICE r5), ASVD_DEBUG
    # every launch; lowrank_check(work) reads it on demand (SVDLinear.check_fused_forward: call it at the end of a generate / eval loop).
    key = work.data_ptr()
    n = _LOWRANK_CALLS.get(key, 0) + 1
    _LOWRANK_CALLS[key] = n
    if os.environ.get("ASVD_DEBUG") or (os.environ.get("ASVD_STRICT") == "1" and n >= 64):
        lowrank_check(work)
    return y


def fro_norm_sq(w):
    lib = L.load(True)
    _dev(w, "w")
    assert w.dim() == 2 and w.stride(1) == 1
    m, n = w.shape
    nb = ctypes.c_size_t()
    L.check(lib.asvd_fro_worksize(m, n, ctypes.byref(nb)), "asvd_fro_worksize")
    work = _work(nb.value, w.device)
    out = torch.empty(1, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        L.check(lib.asvd_fro_norm_sq(_ptr(w), _dt(w), m, n, w.stride(0), _ptr(out), _ptr(work), work.numel(), _stream(w)),
                "asvd_fro_norm_sq")
    return out


def reconstruct_err(W, A, B):
    """(|W - A B|_F^2, |W|_F^2) as a device double[2] tensor"""
    lib = L.load(True)
    _dev(W, "W")
    m, n = W.shape
    r = A.shape[1]
    assert A.shape == (m, r) and B.shape == (r, n) and A.is_contiguous() and B.is_contiguous() and A.dtype == B.dtype
    nb = ctypes.c_size_t()
    L.check(lib.asvd_reconstruct_worksize(m, n, r, ctypes.byref(nb)), "asvd_reconstruct_worksize")
    work = _work(nb.value, W.device)
    out = torch.empty(2, dtype=torch.float64, device=W.device)
    with torch.cuda.device(W.device):
        L.check(lib.asvd_reconstruct_err(_ptr(W), _dt(W), W.stride(0), _ptr(A), _ptr(B), _dt(A), m, n, r, _ptr(out), _ptr(work),
                                         work.numel(), _stream(W)), "asvd_reconstruct_err")
    return out


PROFILE_CLASSES = ["pack", "sgram", "evd", "supdate", "finalize", "snapshot", "gram1", "update1", "supgram"]


def svd_profile(enable=None, keep_split=False):
    """enable/disable HIP-event timing of the SVD kernel classes, or read the last call's totals.  keep_split: the profiled call keeps the
    two-halves split (asvd_svd_set_profiling mode 2; read the halves with svd_split_profile) instead of running every kernel alone on the chip"""
    lib = L.load(False)
    if enable is not None:
        lib.asvd_svd_set_profiling((2 if keep_split else 1) if enable else 0)
        return None
    names = PROFILE_CLASSES
    ms = (ctypes.c_float * len(names))()
    n = (ctypes.c_int * len(names))()
    lib.asvd_svd_get_profile(ms, n)
    out = {names[i]: {"ms": ms[i], "launches": n[i]} for i in range(len(names))}
    cnt = (ctypes.c_longlong * 3)()
    lib.asvd_svd_get_pair_counts(cnt)
    out["pairs"] = {"visited": int(cnt[0]), "rotated": int(cnt[1]), "super_updates": int(cnt[2])}
    sms = (ctypes.c_float * 64)()
    srot = (ctypes.c_longlong * 64)()
    ns = min(64, lib.asvd_svd_get_sweep_times(sms, srot, 64))
    out["sweep_ms"] = [float(sms[i]) for i in range(ns)]
    out["sweep_rotated"] = [int(srot[i]) for i in range(ns)]
    return out


def svd_split_profile():
    """per-half class times of the last call profiled with keep_split (asvd_svd_get_split_profile), or None when that call did not split:
    {"halves": [{class: {"ms", "launches"}} x 2], "supgram_ms": {"half0", "half1", "union", "both"}}"""
    lib = L.load(False)
    n = len(PROFILE_CLASSES)
    ms = (ctypes.c_float * (2 * n))()
    ln = (ctypes.c_int * (2 * n))()
    ov = (ctypes.c_float * 4)()
    if lib.asvd_svd_get_split_profile(ms, ln, ov) != 1:
        return None
    halves = [{PROFILE_CLASSES[i]: {"ms": ms[h * n + i], "launches": ln[h * n + i]} for i in range(n)} for h in range(2)]
    return {"halves": halves, "supgram_ms": {"half0": ov[0], "half1": ov[1], "union": ov[2], "both": ov[3]}}


def svd_set_split(mode):
    """asvd_svd_set_split: 0 never split a batch over the two chip halves, 1 split when allowed, -1 default (as 1 unless ASVD_SPLIT=0)"""
    L.load(False).asvd_svd_set_split(int(mode))
