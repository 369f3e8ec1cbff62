"""Sensitivity sweep — MI355X implementation of sensitivity.py:10-110 behind the same signatures.

calib_sensitivity_ppl:  for every nn.Linear (reverse-DFS order, lm_head first) x candidate ratios, swap in the rank-r
SVDLinear and measure calibration perplexity.  Differences from the reference that do not change results' meaning:
  * the SVD of a layer is computed ONCE (exact, on device) and sliced for all 6 (19 in kv mode) ratios;
  * perplexities are evaluated from cached block inputs of the unmodified model (sweep_eval.py) — bit-identical values;
  * with torch.distributed initialised, layers are LPT-sharded over ranks and the per-layer results are exchanged with one
    all-gather (asvd4llm_amd/parallel.py); every rank returns the complete dict in the reference's insertion order.
calib_sensitivity_stable_rank: forward-free metric -(|W|_F / sigma_max) * ratio**0.1 on the UNSCALED weight; sigma_max
comes from the Lanczos kernel asvd_sigma_max_batched (the reference factorises the whole matrix for this one number),
|W|_F^2 from asvd_fro_norm_sq."""
import os

import torch
import torch.nn as nn
from tqdm import tqdm

from . import ops, parallel
from .evaluate_utils import evaluate_perplexity
from .modules.svd_linear import SVDLinear
from .sweep_eval import PrefixCachedEvaluator


def collect_linear_info(model):
    """{nn.Linear: {"father", "name", "full_name"}} in the order the reference's sweep and search visit the layers (sensitivity.py:19-33,
    binary_search.py:11-27): a module's own Linear children in registration order, then its other children LAST-FIRST, depth first —
    lm_head, then the decoder layers from the last to the first, mlp before self_attn.  tests/golden/linear_order_hf.json pins it."""
    qualified = {module: name for name, module in model.named_modules()}
    found = {}

    def visit(parent):
        containers = []
        for child_name, child in parent.named_children():
            if isinstance(child, nn.Linear):
                found[child] = {"father": parent, "name": child_name, "full_name": qualified[child]}
            else:
                containers.append(child)
        for child in reversed(containers):
            visit(child)

    visit(model)
    return found


def _multi_rank_module(raw_linear, ratios, args):
    """MultiRankSVDLinear for the candidate ratios of one Linear, or None when the fused path does not apply (a rank outside
    [1, min(in, out)] or a failed / NaN factorisation: the per-ratio path then reproduces the reference's fallback behaviour)."""
    from .sweep_eval import MultiRankSVDLinear
    ranks = [SVDLinear.compute_rank(raw_linear, r, args.rank_align) for r in ratios]
    kmax = min(raw_linear.in_features, raw_linear.out_features)
    if min(ranks) < 1 or max(ranks) > kmax:
        return None
    try:
        U, S, V, s = SVDLinear.factorize(raw_linear, True, args.alpha, k=max(ranks), k_compute=getattr(raw_linear, "_asvd_rank_hint", None))
    except Exception:
        return None
    A, B, flags = ops.truncate_split(U, S, V, s, max(ranks), "UV", raw_linear.weight.dtype)  # sweep: from_linear's default sigma_fuse
    if any(int(x) for x in flags.tolist()):
        return None
    bias = raw_linear.bias.data if raw_linear.bias is not None else None
    return MultiRankSVDLinear(A, B, bias, ranks)


def _ppl_candidates(args):
    if args.compress_kv_cache:
        return [0.1 * i for i in range(1, 20)]
    return [0.4, 0.5, 0.6, 0.7, 0.8, 0.9]


@torch.no_grad()
def calib_sensitivity_ppl(model, calib_loader, args, use_cache=True):
    model_id = model.config._name_or_path
    cache_file = f"cache/{model_id.replace('/','_')}_sensitivity_{args.scaling_method}_{args.alpha}_{args.n_calib_samples}_{args.calib_dataset}.pt"
    if use_cache and parallel.cache_exists(cache_file):
        sensitivity_dict = parallel.load_cache(cache_file)
        return sensitivity_dict
    model.eval()
    linear_info = collect_linear_info(model)
    param_ratio_candidates = _ppl_candidates(args)
    input_ids = torch.cat([_["input_ids"] for _ in calib_loader], 0)
    print(f"input_ids.shape={input_ids.shape}")

    rank, ws = parallel.world()
    linears = list(linear_info.items())
    names = [info["full_name"] for _, info in linears]
    # the shard is balanced on the stage that costs the time: the suffix forwards behind each layer (parallel.sweep_layer_costs: a layer of
    # block 0 replays the whole model for every sample, lm_head replays nothing) plus its factorisation — not on the SVD flops alone
    costs = parallel.sweep_costs_for_model(model, linears, len(param_ratio_candidates), min(input_ids.shape[0], args.n_calib_samples), input_ids.shape[1],
                                           prefix_cached=getattr(args, "fused_sweep", True))
    owner = parallel.lpt_assign(costs, ws)
    model._asvd_sweep_owner = {n: o for n, o in zip(names, owner)}   # the factor caches live with these owners: the final decomposition follows them
    model._asvd_sweep_balance = {"predicted_seconds_per_rank_max_over_mean": parallel.load_balance(costs, owner, ws), "world_size": ws}
    keep_cache = getattr(args, "keep_svd_cache", True)
    # what the final decomposition needs to know before it follows this map (binary_search._decompose): the factorisations stay cached with their
    # owners only under keep_svd_cache, and only for these arguments — the same values on every rank, so every rank decides alike
    model._asvd_sweep_owner_meta = {"factors_kept": bool(keep_cache), "alpha": args.alpha, "scaling_method": args.scaling_method, "world_size": ws}

    local = {}
    n_mine = sum(1 for o in owner if o == rank)
    # one factorisation per layer serves every candidate ratio (and the final decomposition): record the largest rank needed
    # and factorise this rank's layers up front, same-shape layers batched (keeps the GPU full; nothing is recomputed later)
    mine = [l for (l, _), o in zip(linears, owner) if o == rank]
    for l in mine:
        l._asvd_rank_hint = max(SVDLinear.compute_rank(l, r, args.rank_align) for r in param_ratio_candidates)
    if keep_cache and getattr(args, "prefactorize", True) and all(l.weight.is_cuda for l in mine):
        SVDLinear.prefactorize(mine, act_aware=True, alpha=args.alpha, ranks={l: l._asvd_rank_hint for l in mine},
                               max_batch=getattr(args, "svd_batch", 32))
    # prefix-cached evaluation (sweep_eval.py): same perplexities, about half the forward work; --no_fused_sweep disables
    evaluator = None
    if getattr(args, "fused_sweep", True) and n_mine > 0:
        evaluator = PrefixCachedEvaluator(model, input_ids, args.n_calib_samples)
    pbar = tqdm(total=n_mine * len(param_ratio_candidates), disable=(rank != 0))
    fused_ratios = evaluator is not None and getattr(args, "fused_ratios", True)
    for (raw_linear, info), own in zip(linears, owner):
        if own != rank:
            continue
        local[info["full_name"]] = {}
        done_ratios = False
        if fused_ratios:
            # all candidate ratios of this layer in ONE batched suffix pass (sweep_eval.MultiRankSVDLinear): one factorisation, the
            # factors at the largest rank, nested truncations as a batch dimension
            multi = _multi_rank_module(raw_linear, param_ratio_candidates, args)
            if multi is not None:
                setattr(info["father"], info["name"], multi)
                try:
                    ppls = evaluator.perplexities(info["full_name"], multi, samples_per_pass=getattr(args, "sweep_samples_per_pass", 1))
                except (RuntimeError, AssertionError, ValueError, TypeError) as e:
                    # a model that rejects the R-fold batch (attention-mask batch checks of older architectures, out of memory on the
                    # R x T x V logits of the 19-ratio kv sweep): the evaluator has restored its hooks and blocks; use the per-ratio path
                    print(f"batched-ratio pass failed for {info['full_name']} ({type(e).__name__}: {e}); evaluating its ratios one by one")
                    if torch.cuda.is_available():
                        torch.cuda.empty_cache()
                    ppls = None
                if ppls is not None:
                    for param_ratio, ppl in zip(param_ratio_candidates, ppls):
                        local[info["full_name"]][param_ratio] = ppl
                        print(f"{info['full_name']} {param_ratio} {ppl}")
                        pbar.update(1)
                    done_ratios = True
        for param_ratio in ([] if done_ratios else param_ratio_candidates):
            svd_linear = SVDLinear.from_linear(
                raw_linear,
                param_ratio=param_ratio,
                alpha=args.alpha,
                act_aware=True,  # hard-coded in the reference sweep (sensitivity.py:50)
                rank_align=args.rank_align,
            )
            setattr(info["father"], info["name"], svd_linear)
            if evaluator is not None:
                ppl = evaluator.perplexity(info["full_name"], svd_linear)
            else:
                ppl = evaluate_perplexity(model, input_ids, args.n_calib_samples)
            local[info["full_name"]][param_ratio] = ppl
            print(f"{info['full_name']} {param_ratio} {ppl}")
            pbar.update(1)
        setattr(info["father"], info["name"], raw_linear)
        if not keep_cache:
            SVDLinear.drop_factor_cache(raw_linear)
    sensitivity_dict = parallel.allgather_sensitivities(local, names, param_ratio_candidates, owner)
    parallel.save_cache(sensitivity_dict, cache_file)
    return sensitivity_dict


@torch.no_grad()
def calib_sensitivity_stable_rank(model, calib_loader, args, use_cache=True):
    model_id = model.config._name_or_path
    cache_file = f"cache/{model_id.replace('/','_')}_sensitivity_stable_rank_{args.scaling_method}_{args.alpha}_{args.n_calib_samples}_{args.calib_dataset}.pt"
    if use_cache and parallel.cache_exists(cache_file):
        sensitivity_dict = parallel.load_cache(cache_file)
        return sensitivity_dict
    model.eval()
    linear_info = collect_linear_info(model)
    param_ratio_candidates = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
    input_ids = torch.cat([_["input_ids"] for _ in calib_loader], 0)
    print(f"input_ids.shape={input_ids.shape}")

    rank, ws = parallel.world()
    linears = list(linear_info.items())
    names = [info["full_name"] for _, info in linears]
    owner = parallel.lpt_assign([parallel.svd_flops(l.out_features, l.in_features) for l, _ in linears], ws)
    local = {}
    pbar = tqdm(total=len(linears) * len(param_ratio_candidates), disable=(rank != 0))
    # sigma_max of this rank's layers: Lanczos kernel (asvd_sigma_max_batched), same-shape layers batched (keeps the GPU full)
    mine = [(raw_linear, info) for (raw_linear, info), own in zip(linears, owner) if own == rank]
    contiguous = {l: (l.weight.data if l.weight.data.stride(1) == 1 else l.weight.data.contiguous()) for l, _ in mine}
    groups = {}
    for l, _ in mine:
        w = contiguous[l]
        groups.setdefault((tuple(w.shape), w.dtype, w.stride(0), w.device), []).append(l)
    sigma_max = {}
    max_batch = getattr(args, "svd_batch", 32)
    for members in groups.values():
        for i in range(0, len(members), max_batch):
            chunk = members[i:i + max_batch]
            sig, _ = ops.sigma_max_batched([contiguous[l] for l in chunk])
            for l, sv in zip(chunk, sig):
                sigma_max[l] = sv
    for raw_linear, info in mine:
        # stable rank = |W|_F / sigma_max on the unscaled weight (sensitivity.py:96-104; scaling commented out there).
        w = raw_linear.weight.data
        wc = contiguous[raw_linear]
        sumsq = ops.fro_norm_sq(wc)  # fp32 sum of squares
        # torch.norm(w, "fro") ** 2 is evaluated in the weight dtype: sqrt rounded to w.dtype, then squared in w.dtype
        w_fro = (sumsq.sqrt().to(w.dtype)) ** 2
        spectral_norm = sigma_max[raw_linear]
        w_spec = spectral_norm ** 2
        sr = (w_fro / w_spec) ** 0.5
        sr = sr.reshape(())
        local[info["full_name"]] = {}
        for param_ratio in param_ratio_candidates:
            local[info["full_name"]][param_ratio] = -sr * param_ratio ** 0.1  # 0-dim tensors, as in the reference
            pbar.update(1)
    if ws > 1:
        as_float = {n: {r: float(v) for r, v in d.items()} for n, d in local.items()}
        full = parallel.allgather_sensitivities(as_float, names, param_ratio_candidates, owner)
        sensitivity_dict = {n: {r: torch.tensor(v) for r, v in d.items()} for n, d in full.items()}
    else:
        sensitivity_dict = {n: local[n] for n in names}
    parallel.save_cache(sensitivity_dict, cache_file)
    return sensitivity_dict
