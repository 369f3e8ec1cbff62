"""Rank allocation + final decomposition — MI355X implementation of binary_search.py:10-131 behind the same signature.

`binary_search_truncation_rank(model, sensitivity_dict, calib_loader, args)` mutates the model in place.  The search is
the reference's (stable sort by -ppl, bisection on the cut index, the last-`mid` slice quirk, identical trace lines); the
decomposition stage slices the cached exact SVD of each layer (computed once by the sweep, or here on first use) instead
of re-factorising, and — with torch.distributed initialised and args.shard_decompose (default) — each rank decomposes only
the layers it owns under the same LPT map as the sweep."""
import time

import torch
import torch.nn as nn
from tqdm import tqdm

from . import parallel
from .evaluate_utils import evaluate_perplexity
from .modules.svd_linear import SVDLinear
from .sensitivity import collect_linear_info


def binary_search_truncation_rank(model, sensitivity_dict, calib_loader, args):
    module_dict = {name: module for name, module in model.named_modules()}
    linear_info = collect_linear_info(model)

    if args.compress_kv_cache:
        ratio_target = args.kv_cache_ratio_target
        sensitivity_dict = {k: v for k, v in sensitivity_dict.items() if "k_proj" in k or "v_proj" in k}
        assert args.ppl_target < 0, "ppl_target is not supported when compressing kv_cache"
        default_param_ratio = 2
    else:
        ratio_target = args.param_ratio_target
        default_param_ratio = 1

    print(
        f"=== {'compress kv_cache' if args.compress_kv_cache else 'compress weight'} target: ppl={args.ppl_target}, ratio_target={ratio_target} ==="
    )

    sensitivity_list = []
    for layername, v in sensitivity_dict.items():
        for param_ratio, ppl in v.items():
            if not args.compress_kv_cache and param_ratio >= 1:
                continue  # weights are to be compressed: ratio must be < 1
            sensitivity_list.append((layername, param_ratio, ppl))
    sorted_sensitive_list = sorted(sensitivity_list, key=lambda x: -x[2])

    high = len(sorted_sensitive_list) - 1
    low = 0
    assert args.ppl_target > 0 or ratio_target > 0

    rank, ws = parallel.world()
    input_ids = torch.cat([_["input_ids"] for _ in calib_loader], 0)
    while low < high:
        mid = (low + high) // 2
        layers_min_ratio = {layername: default_param_ratio for layername in sensitivity_dict.keys()}
        for layername, param_ratio, ppl in sorted_sensitive_list[mid:]:
            layers_min_ratio[layername] = min(layers_min_ratio[layername], param_ratio)
        tot_params = 0
        compress_params = 0
        if args.ppl_target > 0:
            assert not args.compress_kv_cache, "ppl_target is not supported when compressing kv_cache now"
            assert ws == 1, "ppl-target search evaluates the whole model per round: replicas only (SURVEY.md §8e)"
            for layername, param_ratio in layers_min_ratio.items():
                raw_linear = module_dict[layername]
                info = linear_info[raw_linear]
                svd_linear = SVDLinear.from_linear(
                    raw_linear,
                    param_ratio=param_ratio,
                    alpha=args.alpha,
                    act_aware=args.act_aware,
                    sigma_fuse=args.sigma_fuse,
                    rank_align=args.rank_align,
                )
                setattr(info["father"], info["name"], svd_linear)
                tot_params += raw_linear.weight.numel()
                compress_params += raw_linear.weight.numel() * param_ratio
            ppl = evaluate_perplexity(model, input_ids, args.n_calib_samples)
            param_ratio = compress_params / tot_params
            msg = f"low={low} mid={mid}, high={high}, ppl={ppl}, param_ratio={param_ratio}"
            print(msg)
            if ppl < args.ppl_target:
                high = mid
            else:
                low = mid + 1
        else:
            for layername, param_ratio in layers_min_ratio.items():
                raw_linear = module_dict[layername]
                tot_params += raw_linear.weight.numel()
                compress_params += raw_linear.weight.numel() * param_ratio
            now_ratio = compress_params / tot_params
            if args.compress_kv_cache:
                now_ratio /= 2  # param ratio counts ALinear+BLinear, the rank ratio is half of it
            msg = f"low={low} mid={mid}, high={high}, now_ratio={now_ratio}, params=({compress_params}/{tot_params})"
            print(msg)
            if now_ratio > ratio_target:
                high = mid
            else:
                low = mid + 1

    print(f"=== Searching done, decomposing layers... ===")
    layers_min_ratio = {layername: default_param_ratio for layername in sensitivity_dict.keys()}
    for layername, param_ratio, ppl in sorted_sensitive_list[mid:]:  # the LAST mid, not low (reference quirk, :106)
        layers_min_ratio[layername] = min(layers_min_ratio[layername], param_ratio)

    # ownership under the same LPT map as the sweep (all Linears in traversal order)
    shard = ws > 1 and getattr(args, "shard_decompose", True)
    linears = list(linear_info.items())
    owner_list = parallel.lpt_assign([parallel.svd_flops(l.out_features, l.in_features) for l, _ in linears], ws)
    owner = {info["full_name"]: o for (_, info), o in zip(linears, owner_list)}

    # tied weights (OPT: lm_head.weight IS embed_tokens.weight): the reference's `raw_linear.to("cpu")` would drag the embedding to the
    # CPU with it and break the next forward; only offload weights no other module shares
    uses = {}
    for _, prm in model.named_parameters(remove_duplicate=False):
        uses[id(prm)] = uses.get(id(prm), 0) + 1

    st = time.time()
    exchange_items = []
    for layername, param_ratio in tqdm(layers_min_ratio.items(), disable=(rank != 0)):
        raw_linear = module_dict[layername]
        info = linear_info[raw_linear]
        if param_ratio == default_param_ratio:
            svd_linear = raw_linear
        elif shard and owner[layername] != rank:
            svd_linear = raw_linear  # another rank owns this layer's factors: they arrive in exchange_factors below
            exchange_items.append((layername, info["father"], info["name"], raw_linear))
        else:
            exchange_items.append((layername, info["father"], info["name"], raw_linear))
            svd_linear = SVDLinear.from_linear(
                raw_linear,
                param_ratio=param_ratio,
                alpha=args.alpha,
                act_aware=args.act_aware,
                sigma_fuse=args.sigma_fuse,
                rank_align=args.rank_align,
            )
            SVDLinear.drop_factor_cache(raw_linear)
            if getattr(args, "offload_raw_to_cpu", True) and uses.get(id(raw_linear.weight), 1) <= 1:
                raw_linear.to("cpu")  # binary_search.py:127
        setattr(info["father"], info["name"], svd_linear)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    ed = time.time()
    print(f"decompose time: {ed-st}")
    if shard:
        # complete the model: every owner hands its factors to rank 0 (default) or to everybody — without this only ~1/ws of the
        # layers of any one replica would be factorised and an export / evaluation would silently miss the target
        mode = getattr(args, "gather_factors", "rank0")
        t0 = time.time()
        got = parallel.exchange_factors(exchange_items, owner, mode)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        model._asvd_factor_exchange = {"mode": mode, "received": got, "seconds": time.time() - t0}
        if rank == 0:
            print(f"factor exchange ({mode}): received {got} layers in {time.time() - t0:.2f} s")
    model._asvd_layers_min_ratio = layers_min_ratio
    model._asvd_decompose_time = ed - st
