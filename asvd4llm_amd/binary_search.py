"""Rank allocation + final decomposition behind the reference's entry point (binary_search.py:10-131):

    binary_search_truncation_rank(model, sensitivity_dict, calib_loader, args)      # mutates the model in place

What the reference computes — kept, because the fixtures generated from it pin the trace lines and the final ranks:
  * candidates (layer, ratio, ppl), ratio < 1 unless compressing the kv cache, stably sorted by descending ppl;
  * a bisection on the cut index `c` of that list: the plan of a cut gives every layer the smallest ratio among its candidates
    at positions >= c (or the default 1 / 2 when it has none there); the plan's parameter ratio (or, with --ppl_target, the
    perplexity of the model compressed to the plan) decides the half;
  * the decomposition uses the plan of the LAST PROBED cut, not of the converged bound (binary_search.py:106).

How it is done here (own design):
  * `_CutPlans` indexes the sorted list once: per layer the positions of its candidates with the running minimum of their ratios
    from the back, so the plan of any cut is one bisect per layer — O(#layers log #ratios) per probe instead of a scan of the
    whole list per probe (1350 entries x 11 probes for Llama-2-7B);
  * the bisection is a generic `_bisect_cut(n, probe)`; the two targets are two probe closures that also print the trace line;
  * the decomposition stage slices the cached exact SVD of each layer (computed once by the sweep, or here on first use) instead
    of re-factorising, and — with torch.distributed initialised and args.shard_decompose (default) — each rank decomposes only
    the layers it owns under the same LPT map as the sweep, then the factors are exchanged (parallel.exchange_factors)."""
import bisect
import time

import torch
from tqdm import tqdm

from . import parallel
from .evaluate_utils import evaluate_perplexity
from .modules.svd_linear import SVDLinear
from .sensitivity import collect_linear_info


class _CutPlans:
    """The sorted candidate list, indexed by layer for suffix minima."""

    def __init__(self, sensitivity, keep_ratio, default_ratio):
        self.layers = list(sensitivity.keys())  # plan order = insertion order of the sensitivity dict (the parameter sums run in it)
        self.default = default_ratio
        cand = [(name, ratio, ppl) for name, per_ratio in sensitivity.items() for ratio, ppl in per_ratio.items() if keep_ratio(ratio)]
        order = sorted(range(len(cand)), key=lambda i: -cand[i][2])  # stable: ties keep the traversal order
        self.size = len(order)
        pos = {name: [] for name in self.layers}
        for p, i in enumerate(order):
            pos[cand[i][0]].append((p, cand[i][1]))
        self._pos, self._sufmin = {}, {}
        for name, pr in pos.items():  # pr is ascending in position
            self._pos[name] = [p for p, _ in pr]
            mins, cur = [], None
            for _, ratio in reversed(pr):
                cur = ratio if cur is None else min(cur, ratio)
                mins.append(cur)
            self._sufmin[name] = mins[::-1]

    def plan(self, cut):
        """{layer: smallest candidate ratio at sorted positions >= cut, else the default} in plan order"""
        out = {}
        for name in self.layers:
            j = bisect.bisect_left(self._pos[name], cut)
            out[name] = min(self.default, self._sufmin[name][j]) if j < len(self._pos[name]) else self.default
        return out


def _bisect_cut(n, probe):
    """The reference's bisection over cut indices 0 .. n-1: probe(low, mid, high) -> True keeps the lower half.
    Returns the last probed index (None when n < 2: nothing was probed)."""
    low, high, last = 0, n - 1, None
    while low < high:
        last = (low + high) // 2
        if probe(low, last, high):
            high = last
        else:
            low = last + 1
    return last


def _plan_params(plan, weights):
    """(compressed, total) parameter counts of a plan, accumulated layer by layer in plan order (floats: the trace prints them)"""
    total, compressed = 0, 0
    for name, ratio in plan.items():
        n = weights[name]
        total += n
        compressed += n * ratio
    return compressed, total


def binary_search_truncation_rank(model, sensitivity_dict, calib_loader, args):
    modules = dict(model.named_modules())
    linear_info = collect_linear_info(model)

    kv = bool(args.compress_kv_cache)
    if kv:
        assert args.ppl_target < 0, "ppl_target is not supported when compressing kv_cache"
        sensitivity_dict = {k: v for k, v in sensitivity_dict.items() if "k_proj" in k or "v_proj" in k}
    ratio_target = args.kv_cache_ratio_target if kv else args.param_ratio_target
    default_ratio = 2 if kv else 1
    print(f"=== {'compress kv_cache' if kv else 'compress weight'} target: ppl={args.ppl_target}, ratio_target={ratio_target} ===")
    assert args.ppl_target > 0 or ratio_target > 0

    # weights are to be compressed: only ratios < 1 are candidates; the kv-cache sweep keeps all of its 0.1 .. 1.9
    plans = _CutPlans(sensitivity_dict, (lambda r: True) if kv else (lambda r: r < 1), default_ratio)
    weights = {name: modules[name].weight.numel() for name in plans.layers}
    rank, ws = parallel.world()
    from_linear_kw = dict(alpha=args.alpha, act_aware=args.act_aware, sigma_fuse=args.sigma_fuse, rank_align=args.rank_align)

    if args.ppl_target > 0:
        assert ws == 1, "ppl-target search evaluates the whole model per round: replicas only (SURVEY.md §8e)"
        input_ids = torch.cat([_["input_ids"] for _ in calib_loader], 0)

        def probe(low, mid, high):
            plan = plans.plan(mid)
            for name, ratio in plan.items():  # the whole model compressed to this plan (every layer, ratio 1 included)
                raw = modules[name]
                info = linear_info[raw]
                setattr(info["father"], info["name"], SVDLinear.from_linear(raw, param_ratio=ratio, **from_linear_kw))
            ppl = evaluate_perplexity(model, input_ids, args.n_calib_samples)
            compressed, total = _plan_params(plan, weights)
            print(f"low={low} mid={mid}, high={high}, ppl={ppl}, param_ratio={compressed / total}")
            return ppl < args.ppl_target
    else:
        def probe(low, mid, high):
            compressed, total = _plan_params(plans.plan(mid), weights)
            now_ratio = compressed / total
            if kv:
                now_ratio /= 2  # the ratio counts ALinear + BLinear; the kv-cache (rank) ratio is half of it
            print(f"low={low} mid={mid}, high={high}, now_ratio={now_ratio}, params=({compressed}/{total})")
            return now_ratio > ratio_target

    last_cut = _bisect_cut(plans.size, probe)
    print("=== Searching done, decomposing layers... ===")
    if last_cut is None:  # fewer than two candidates: the reference dies on its unbound `mid` here (binary_search.py:106)
        raise UnboundLocalError("binary search needs at least two (layer, ratio) candidates: no cut was probed")
    layers_min_ratio = plans.plan(last_cut)  # the LAST PROBED cut, not the converged one (reference quirk)

    _decompose(model, modules, linear_info, layers_min_ratio, default_ratio, from_linear_kw, args, rank, ws)
    model._asvd_layers_min_ratio = layers_min_ratio


def _decompose_owner_map(model, linear_info, args, ws, shard):
    """{full_name: owning rank} of the final decomposition — identical on every rank."""
    # ownership under the same LPT map as the sweep (all Linears in traversal order)
    linears = list(linear_info.items())
    owner_list = parallel.lpt_assign([parallel.svd_flops(l.out_features, l.in_features) for l, _ in linears], ws)
    owner = {info["full_name"]: o for (_, info), o in zip(linears, owner_list)}
    # a ppl sweep that ran in this process sharded the layers on ITS cost model (parallel.sweep_layer_costs) and left the factorisations cached
    # with those owners: follow that map (slicing a cached factorisation is ~1 ms; the map above would re-factorise on another rank).  The map
    # is a deterministic function of the model and the arguments, so it is identical on every rank; a sensitivity dict that came from a
    # cache file or from the stable-rank metric has no such map and the decomposition is balanced on its own cost, the SVD flops.
    # Followed ONLY while those caches exist: the sweep's map is balanced on suffix-forward seconds — ranks that own late-block layers hold
    # many more layers — so with keep_svd_cache off, or on a model object whose arguments changed since the sweep, every layer would be re-factorised
    # under a shard that is badly unbalanced for SVD flops.  The test uses what the sweep recorded (identical on every rank), not rank-local state.
    sweep_owner = getattr(model, "_asvd_sweep_owner", None)
    meta = getattr(model, "_asvd_sweep_owner_meta", None) or {}
    caches_live = (meta.get("factors_kept") is True and meta.get("alpha") == getattr(args, "alpha", None)
                   and meta.get("scaling_method") == getattr(args, "scaling_method", None) and meta.get("world_size") == ws)
    if shard and sweep_owner is not None and caches_live and set(sweep_owner) == set(owner) and max(sweep_owner.values()) < ws:
        owner = dict(sweep_owner)
    return owner


def _decompose(model, modules, linear_info, layers_min_ratio, default_ratio, from_linear_kw, args, rank, ws):
    """Swap in the factorised layers of the chosen plan (reference: binary_search.py:111-131, which also times this stage)."""
    shard = ws > 1 and getattr(args, "shard_decompose", True)
    owner = _decompose_owner_map(model, linear_info, args, ws, shard)
    # tied weights (OPT: lm_head.weight IS embed_tokens.weight): the reference's `raw_linear.to("cpu")` would drag the embedding to the
    # CPU with it and break the next forward; only weights no other module shares are offloaded
    uses = {}
    for _, prm in model.named_parameters(remove_duplicate=False):
        uses[id(prm)] = uses.get(id(prm), 0) + 1
    offload = getattr(args, "offload_raw_to_cpu", True)

    t_start = time.time()
    selected = []  # (full_name, father, child name, raw Linear) of every layer the plan factorises: what exchange_factors walks
    for name, ratio in tqdm(layers_min_ratio.items(), disable=(rank != 0)):
        raw = modules[name]
        info = linear_info[raw]
        if ratio == default_ratio:
            setattr(info["father"], info["name"], raw)  # not selected: the raw Linear (a ppl-target probe may have left a module in its place)
            continue
        selected.append((name, info["father"], info["name"], raw))
        if shard and owner[name] != rank:
            continue  # another rank owns this layer's factors: they arrive in exchange_factors below
        setattr(info["father"], info["name"], SVDLinear.from_linear(raw, param_ratio=ratio, **from_linear_kw))
        SVDLinear.drop_factor_cache(raw)
        if offload and uses.get(id(raw.weight), 1) <= 1:
            raw.to("cpu")  # binary_search.py:127
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t_end = time.time()
    print(f"decompose time: {t_end-t_start}")
    model._asvd_decompose_time = t_end - t_start
    if shard:
        # complete the model: every owner hands its factors to rank 0 (default) or to everybody — without this only ~1/ws of the
        # layers of any one replica would be factorised and an export / evaluation would silently miss the target
        mode = getattr(args, "gather_factors", "rank0")
        t0 = time.time()
        got = parallel.exchange_factors(selected, owner, mode)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        model._asvd_factor_exchange = {"mode": mode, "received": got, "seconds": time.time() - t0}
        if rank == 0:
            print(f"factor exchange ({mode}): received {got} layers in {time.time() - t0:.2f} s")
