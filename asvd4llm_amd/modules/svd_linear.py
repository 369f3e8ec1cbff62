"""SVDLinear / SVDLinear.from_linear — MI355X implementation behind the reference API (modules/svd_linear.py:7-109).

Same class, same constructor, same `from_linear` signature, same observable failure behaviour (printed messages and a
freshly initialised nn.Linear, svd_linear.py:66-68,81-98).  The bodies call libasvd_hip.so through asvd4llm_amd.ops:

  * ONE exact SVD of W*diag(s) per nn.Linear, kept on the device (288 GB HBM) and sliced for every candidate rank.  The
    reference re-factorises (randomised torch.svd_lowrank) for each of the 6..19 ratios of the sweep and again for the
    final decomposition (sensitivity.py:43-52, binary_search.py:119-126).
  * truncation, un-scaling by s, sigma fusion, transpose and the cast to the weight dtype are one fused kernel pair.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import AsvdHipError

# set ASVD_STRICT=1 (benchmarks / tests do) to raise instead of silently substituting a random Linear
def _strict():
    return os.environ.get("ASVD_STRICT", "0") == "1"


FUSED_FORWARD_MAX_TOKENS, FUSED_FORWARD_MAX_OUT = 16, 8192  # where the one-launch forward measured faster than two GEMM launches (tokens; in/out features)


def _fused_forward_enabled():
    """The one-launch forward is OPT-IN (ASVD_FUSED_FORWARD=1, or `module.fused_forward = True` on one SVDLinear): it runs from padded
    COPIES of the two factors, so it must only be used on weights that are no longer edited through `.data` (such edits do not bump
    Parameter._version and cannot be seen without reading the weights; call `refresh_fused()` after one)."""
    return os.environ.get("ASVD_FUSED_FORWARD", "0") == "1"


def _plain_linear_without_hooks(mod):
    """the fused launch replaces ALinear.forward / BLinear.forward: only legitimate when those are exactly nn.Linear.forward with nothing
    hooked on (calibration hooks of a re-calibrated compressed model, wrappers, subclasses, global hooks must keep running)"""
    import torch.nn.modules.module as _m
    return (type(mod) is nn.Linear and not mod._forward_hooks and not mod._forward_pre_hooks and not mod._backward_hooks
            and not _m._global_forward_hooks and not _m._global_forward_pre_hooks)


class SVDLinear(nn.Module):
    """nn.Module{ALinear: Linear(r->out, bias?), BLinear: Linear(in->r, no bias), truncation_rank} (svd_linear.py:7-24)."""

    def __init__(self, U, S, V, bias=None, sigma_fuse="UV") -> None:
        super().__init__()
        self.ALinear = nn.Linear(U.size(1), U.size(0), bias=bias is not None)
        if bias is not None:
            self.ALinear.bias.data = bias
        self.BLinear = nn.Linear(V.size(1), V.size(0), bias=False)
        self.truncation_rank = S.size(0)
        if U.is_cuda:
            A, B, _ = ops.truncate_split(U.float(), S.float(), V.float(), None, S.size(0), sigma_fuse, torch.float32)
            self.ALinear.weight.data = A
            self.BLinear.weight.data = B
        else:
            raise AsvdHipError("SVDLinear(U, S, V) needs device tensors: the ASVD hot path has no CPU fallback")

    @classmethod
    def _from_factors(cls, A, B, bias, rank):
        """build from already fused/cast factors (the product path: no temporary fp32 Linear weights)"""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        with torch.device("meta"):
            a = nn.Linear(A.shape[1], A.shape[0], bias=bias is not None)
            b = nn.Linear(B.shape[1], B.shape[0], bias=False)
        a.weight = nn.Parameter(A, requires_grad=True)
        b.weight = nn.Parameter(B, requires_grad=True)
        if bias is not None:
            a.bias = nn.Parameter(bias, requires_grad=True)
        self.ALinear, self.BLinear = a, b
        self.truncation_rank = rank
        return self

    # ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def compute_rank(linear, param_ratio, rank_align=1):
        """svd_linear.py:39-44"""
        n_params = linear.weight.numel()
        compressed_params = int(n_params * param_ratio)
        rank = compressed_params // (linear.in_features + linear.out_features)
        rank = int(np.ceil(rank / rank_align) * rank_align)
        return rank

    @staticmethod
    def _scale_vector(linear, act_aware, alpha):
        """svd_linear.py:48-59: s = 1 * scaling**alpha [* fisher**alpha] + 1e-6, in the dtype of the statistics"""
        if not act_aware:
            return None
        scaling = getattr(linear, "scaling_diag_matrix", None)
        fisher = getattr(linear, "fisher_info", None)
        if scaling is None and fisher is None:
            # the reference would crash on `float.view` here (svd_linear.py:60); keep that loud
            raise AttributeError("act_aware=True but the Linear has neither scaling_diag_matrix nor fisher_info")
        dev = linear.weight.device
        if scaling is None:
            scaling, fisher = fisher, None
        scaling = scaling.to(dev)
        return ops.make_scale(scaling, None if fisher is None else fisher.to(dev), alpha=alpha, eps=1e-6)

    @staticmethod
    def _cache_key(linear, act_aware, alpha):
        """(key, contiguous weight).  Content signature: `.data` edits do not bump Parameter._version, so the values are hashed:
        sum of squares of every element (K8 kernel) AND a position-weighted fp64 checksum over a strided sample of <= 8192 elements
        (sign flips and permutations keep the first and change the second), plus Parameter._version — all read back in ONE host sync."""
        w = linear.weight.data
        if not w.is_cuda:
            raise AsvdHipError(f"weight of {linear} is on {w.device}; the ASVD hot path runs on gfx950 only (no CPU fallback)")
        stat = getattr(linear, "scaling_diag_matrix", None) if act_aware else None
        fis = getattr(linear, "fisher_info", None) if act_aware else None
        wc = w if w.stride(1) == 1 else w.contiguous()
        pend = []  # device scalars (fp64) to read back together

        def _sig(t):
            if t is None:
                return None
            t2 = t.to(w.device).reshape(1, -1) if t.dim() != 2 else t
            t2 = t2 if t2.stride(-1) == 1 else t2.contiguous()
            flat = t2.reshape(-1) if t2.is_contiguous() else t2.contiguous().reshape(-1)
            step = max(1, flat.numel() // 8192)
            samp = flat[::step].double()
            pend.append(ops.fro_norm_sq(t2).double().reshape(()))
            pend.append((samp * torch.arange(1, samp.numel() + 1, dtype=torch.float64, device=samp.device)).sum())
            return (tuple(t.shape), t.dtype, len(pend) - 2)

        sigs = [_sig(wc), _sig(stat), _sig(fis)]
        vals = torch.stack(pend).tolist()  # the one host sync
        sigs = [None if sg is None else (sg[0], sg[1], vals[sg[2]], vals[sg[2] + 1]) for sg in sigs]
        key = (w.data_ptr(), int(linear.weight._version), sigs[0], bool(act_aware), float(alpha) if act_aware else None, sigs[1], sigs[2])
        return key, wc

    @staticmethod
    def factorize(linear, act_aware=False, alpha=1, k=None, k_compute=None):
        """One exact SVD of W*diag(s) on the device, cached on the module.  Returns (U, S, V, s).
        k: number of leading triplets needed now (None = all); a cached factorisation is reused when it holds >= k columns.
        k_compute: how many to compute on a cache miss (None = all, so that any later rank is a slice of this one SVD)."""
        key, wc = SVDLinear._cache_key(linear, act_aware, alpha)
        kmax = min(wc.shape)
        k = kmax if k is None else max(1, min(int(k), kmax))
        cache = getattr(linear, "_asvd_factor_cache", None)
        if cache is not None and cache[0] == key and cache[1][1].numel() >= k:
            return cache[1]
        k = kmax if k_compute is None else max(k, min(int(k_compute), kmax))
        s = SVDLinear._scale_vector(linear, act_aware, alpha)
        U, S, V, info = ops.svd(wc, s, k=k)
        if info.status == 2:
            raise FloatingPointError("nan in svd")
        if info.status == 1:
            # ASVD_N_NOCONV: the sweeps stopped at max_sweeps.  Never cache an unconverged factorisation silently: one retry with a
            # generous sweep budget, then strict mode raises and the default path warns (results of the retry are still returned).
            U, S, V, info = ops.svd(wc, s, k=k, max_sweeps=60)
            if info.status == 2:
                raise FloatingPointError("nan in svd")
            if info.status == 1:
                if _strict():
                    raise ArithmeticError(f"block-Jacobi SVD of {tuple(wc.shape)} did not converge in 60 sweeps")
                print(f"warning: SVD of {linear} not converged after 60 sweeps (last rotated pairs {info.last_rotated_pairs})")
        linear._asvd_factor_cache = (key, (U, S, V, s))
        linear._asvd_svd_info = info
        return U, S, V, s

    @staticmethod
    def prefactorize(linears, act_aware=False, alpha=1, ranks=None, max_batch=32, concurrent=True):
        """Factorise many Linears up front, same-shape ones CONCURRENTLY (asvd_svd_batched): independent matrices are what
        fills the 256 CUs during the 64-workgroup eigen-solve phase, and 288 GB of HBM hold the factors of a whole 7B/13B
        shard.  ranks: optional {linear: largest rank needed} (convergence is then only enforced for those leading triplets).
        Fills the same per-module cache `from_linear` uses; already cached layers are skipped."""
        groups = {}
        for lin in linears:
            key, wc = SVDLinear._cache_key(lin, act_aware, alpha)
            kmax = min(wc.shape)
            k = kmax if not ranks or not ranks.get(lin) else max(1, min(int(ranks[lin]), kmax))
            cache = getattr(lin, "_asvd_factor_cache", None)
            if cache is not None and cache[0] == key and cache[1][1].numel() >= k:
                continue
            s = SVDLinear._scale_vector(lin, act_aware, alpha)
            gk = (tuple(wc.shape), wc.dtype, wc.stride(0), wc.device, None if s is None else s.dtype)
            groups.setdefault(gk, []).append((lin, key, wc, s, k))
        jobs = []
        for gk, items in groups.items():
            # max_batch is the chunk size of the big shapes (>= 2048 columns: one chunk of 32 fills every phase of a sweep); smaller problems are
            # latency chains — 768 x 768: 944 SVD/s in chunks of 16, 1568 at 32, 1812 at 64, 2270 at 128 (bench.py --m 768 --n 768 --batch ...) —
            # and their workspaces are small: twice / four times the chunk below 2048 / 1024 columns
            kcols = min(gk[0])
            chunk_size = max_batch * (1 if kcols >= 2048 else (2 if kcols >= 1024 else 4))
            for i in range(0, len(items), chunk_size):
                jobs.append((kcols, items[i:i + chunk_size]))

        def run(chunk):
            k = max(it[4] for it in chunk)
            scs = None if chunk[0][3] is None else [it[3] for it in chunk]
            U, S, V, infos = ops.svd_batched([it[2] for it in chunk], scs, k=k)
            for j, (lin, key, wc, s, _) in enumerate(chunk):
                if infos[j].status != 0:
                    continue  # NaN / not converged: from_linear -> factorize handles it (retry, strict raise, or NaN fallback)
                lin._asvd_factor_cache = (key, (U[j], S[j], V[j], s))
                lin._asvd_svd_info = infos[j]
            return U, S, V

        # Calls of small problems (< 2048 columns) leave most of the chip idle between the launches of their dependency chain, and the library
        # takes concurrent calls on different streams (include/asvd_hip.h; tests/test_gpu_concurrency.py): the groups of an opt-125m-shaped model
        # (48 x 768^2, 12 x 3072x768, 12 x 768x3072, the 50272x768 lm_head) run side by side, one host thread and one stream each.
        # Round 5: the same goes for LONE large problems (fewer than 4 of a shape: the lm_head of a Llama — one 32000 x 4096 matrix is a latency chain of
        # ~0.1 s that fills 32 of the 256 CUs): they run on side threads WHILE the caller's thread works through the large batches.
        def side(kc, c):
            return c[0][2].is_cuda and (kc < 2048 or len(c) < 4)

        small = [c for kc, c in jobs if side(kc, c)]
        n_main = sum(1 for kc, c in jobs if not side(kc, c))
        if concurrent and (len(small) > 1 or (small and n_main > 0)):
            import threading
            from .. import _lib
            _lib.load(True)   # loaded (and, if need be, built) once, by this thread
            # every chunk lives on ONE device (the group key holds it): its side stream, the stream it waits for and the stream its outputs
            # are handed back to are that device's.  A small fixed pool of side streams per device is reused by the workers (a fresh stream
            # per chunk would leave its workspace cached in a pool nobody uses again)
            errors, outs = [], []
            lock = threading.Lock()
            nslots = 4
            pools = {}
            for c in small:
                d = c[0][2].device
                if d not in pools:   # (side streams, admission, free slots, the CALLER's current stream of that device — current streams are per thread)
                    pools[d] = ([torch.cuda.Stream(device=d) for _ in range(nslots)], threading.Semaphore(nslots), list(range(nslots)), torch.cuda.current_stream(d))

            def worker(chunk):
                dev = chunk[0][2].device
                streams, sem, free, main = pools[dev]
                with sem:
                    with lock:
                        slot = free.pop()
                    try:
                        st = streams[slot]
                        st.wait_stream(main)   # weights and scale vectors were produced on the caller's stream of that device
                        with torch.cuda.device(dev), torch.cuda.stream(st):
                            res = run(chunk)
                        st.synchronize()
                        with lock:
                            outs.append((dev, res))
                    except Exception as e:  # noqa: BLE001 — re-raised by the caller's thread below
                        with lock:
                            errors.append(e)
                    finally:
                        with lock:
                            free.append(slot)

            threads = [threading.Thread(target=worker, args=(c,)) for c in small]
            for t in threads:
                t.start()
            main_error = None
            try:
                for kc, chunk in jobs:   # the large batches, on the caller's thread and stream, next to the side workers
                    if not side(kc, chunk):
                        run(chunk)
            except Exception as e:  # noqa: BLE001 — the workers are joined before anything is raised
                main_error = e
            for t in threads:
                t.join()
            for dev, (U, S, V) in outs:   # allocated on the side streams, consumed on the caller's stream of their device from here on
                main = pools[dev][3]
                for lst in (U, S, V):
                    for t in (lst or []):
                        t.record_stream(main)
            if main_error is not None:
                raise main_error
            if errors:
                raise errors[0]
            return
        for _, chunk in jobs:
            run(chunk)

    @staticmethod
    def drop_factor_cache(linear):
        if hasattr(linear, "_asvd_factor_cache"):
            del linear._asvd_factor_cache

    @staticmethod
    def from_linear(
        linear: nn.Linear,
        param_ratio: float,
        act_aware=False,
        ic_split=1,
        oc_split=1,
        alpha=1,
        sigma_fuse="UV",
        rank_align=1,
    ):
        assert ic_split == 1 or oc_split == 1
        rank = SVDLinear.compute_rank(linear, param_ratio, rank_align)
        dtype, device = linear.weight.dtype, linear.weight.device

        def fallback():  # svd_linear.py:68,84-98 — a freshly initialised Linear (!), kept for behavioural parity
            return nn.Linear(linear.in_features, linear.out_features).to(dtype).to(device)

        try:
            if rank < 1 or rank > min(linear.in_features, linear.out_features):
                raise ValueError(f"rank {rank} outside [1, min(in, out)]")  # torch.svd_lowrank(q=rank) raises too
            hint = getattr(linear, "_asvd_rank_hint", None)  # set by the sweep: the largest rank it will ask for
            U, S, V, s = SVDLinear.factorize(linear, act_aware, alpha, k=rank, k_compute=hint)
        except (AsvdHipError, AttributeError):
            raise
        except Exception:
            if _strict():
                raise
            print(f"svd failed for {linear}, disable act_aware")
            return fallback()

        A, B, flags = ops.truncate_split(U, S, V, s, rank, sigma_fuse, dtype)
        nan_s, nan_u, nan_v = (int(x) for x in flags.tolist())  # one host sync, as the reference's .any() checks
        for bad, what in ((nan_s, "S"), (nan_u, "U"), (nan_v, "V")):
            if bad:
                if _strict():
                    raise FloatingPointError(f"nan in {what}")
                print(f"nan in {what}")
                return fallback()
        bias = linear.bias.data if linear.bias is not None else None
        new_linear = SVDLinear._from_factors(A, B, bias, rank)
        new_linear.to(dtype)
        return new_linear

    def refresh_fused(self):
        """drop the padded factor copies of the one-launch forward (rebuilt on the next decode-sized call)"""
        self._fused = None

    def check_fused_forward(self):
        """Read (one host sync per workspace) and clear the give-up flag of every one-launch-forward workspace of this module; raises AsvdHipError when
        a launch since the last check left its in-kernel grid barrier (its output was NaN-poisoned).  The natural place is the end of a generate /
        evaluation loop: `for m in model.modules(): isinstance(m, SVDLinear) and m.check_fused_forward()`."""
        st = getattr(self, "_fused", None)
        if st is not None:
            for work in st[3].values():
                ops.lowrank_check(work)

    def _fused_state(self, stream_id):
        """Padded copies of the two factors for the one-launch forward (ops.lowrank_pack), rebuilt when either Parameter is replaced or
        bumps its version, and ONE barrier/intermediate workspace PER STREAM (two launches in flight must not share barrier words)."""
        A, B = self.ALinear.weight, self.BLinear.weight
        key = (A.data_ptr(), A._version, B.data_ptr(), B._version, A.device)
        st = getattr(self, "_fused", None)
        if st is None or st[0] != key:
            Ap, Bp, work = ops.lowrank_pack(A.detach(), B.detach())
            st = (key, Ap, Bp, {stream_id: work})
            self._fused = st
        works = st[3]
        if stream_id not in works:
            works[stream_id] = torch.zeros_like(next(iter(works.values())))
        return st[1], st[2], works[stream_id]

    def forward(self, inp):
        # compute USV^Tx + b  (svd_linear.py:105-109).  Decode-sized inputs (<= 16 fp16 tokens, in/out features <= 8192, no autograd): ONE
        # persistent launch that streams B and A once and keeps the r-wide intermediate on chip (K10, csrc/lowrank_forward.hip) — measured
        # 20.9 / 22.2 / 25.0 / 32.2 us at 1 / 2 / 4 / 16 tokens against 37 us for the two hipBLASLt launches at 4096 -> 1843 -> 4096 (round 5,
        # profiles/r5_k10_nostrict.jsonl; under ASVD_STRICT the wrapper used to read the give-up flag after every launch: +18 us, now every 64th).  Everywhere
        # else the reference's two GEMMs through nn.Linear: at par from 64 tokens on and on the 11008-wide MLP projections (DESIGN.md
        # section 4 has the table).  Opt-in (ASVD_FUSED_FORWARD=1 / self.fused_forward): see _fused_forward_enabled.
        if ((getattr(self, "fused_forward", False) or _fused_forward_enabled()) and inp.is_cuda and inp.dtype == torch.float16 and self.BLinear.weight.dtype == torch.float16 and inp.shape[-1] % 64 == 0
                and 0 < inp.numel() // inp.shape[-1] <= FUSED_FORWARD_MAX_TOKENS and max(self.ALinear.out_features, inp.shape[-1]) <= FUSED_FORWARD_MAX_OUT
                and _plain_linear_without_hooks(self.ALinear) and _plain_linear_without_hooks(self.BLinear) and self.BLinear.bias is None
                and not (torch.is_grad_enabled() and (inp.requires_grad or self.ALinear.weight.requires_grad or self.BLinear.weight.requires_grad))):
            Ap, Bp, work = self._fused_state(torch.cuda.current_stream(inp.device).cuda_stream)
            x2d = inp.reshape(-1, inp.shape[-1])
            y = ops.lowrank_forward(x2d if x2d.is_contiguous() else x2d.contiguous(), Ap, Bp, self.ALinear.bias, work)
            return y.view(*inp.shape[:-1], y.shape[-1])
        y = self.BLinear(inp)
        y = self.ALinear(y)
        return y
