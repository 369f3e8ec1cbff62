"""Full-model decomposition wall-clock (BASELINE.json configs[2]: Llama-2-7b-shaped, alpha 0.5, param_ratio 0.9, all Linears, 1 GPU).

Measures the stage the reference itself times as `decompose time` (binary_search.py:111-131): for every nn.Linear of the model,
scale by the activation statistics, factorise, truncate to the rank of the target ratio, split into ALinear/BLinear (fp16).
Weights and statistics are synthetic (shape-faithful; no checkpoints offline).  The model forwards of the ppl sweep are ordinary
PyTorch execution and are not part of this number."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

SHAPES = {
    "llama-2-7b": dict(layers=32, attn=(4096, 4096), mlp_up=(11008, 4096), mlp_down=(4096, 11008), head=(32000, 4096)),
    "llama-2-13b": dict(layers=40, attn=(5120, 5120), mlp_up=(13824, 5120), mlp_down=(5120, 13824), head=(32000, 5120)),
    "opt-125m": dict(layers=12, attn=(768, 768), mlp_up=(3072, 768), mlp_down=(768, 3072), head=(50272, 768)),
}


def build_linears(name, dev, n_layers=None, seed=233):
    cfg = SHAPES[name]
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    L = n_layers or cfg["layers"]

    def mk(o, i, tag):
        lin = nn.Linear(i, o, bias=False, device="meta")
        w = (torch.randn(o, i, generator=g) * 0.02).half()
        lin.weight = nn.Parameter(w.to(dev), requires_grad=False)
        scal = (32 * torch.randn(i, generator=g).abs())
        k = max(1, int(0.01 * i)); scal[torch.randperm(i, generator=g)[:k]] *= 30
        lin.scaling_diag_matrix = scal.half().to(dev)
        lin._tag = tag
        return lin

    for l in range(L):
        for nm in ("q", "k", "v", "o"):
            out.append(mk(*cfg["attn"], f"l{l}.{nm}"))
        for nm in (("fc1",) if name.startswith("opt") else ("gate", "up")):
            out.append(mk(*cfg["mlp_up"], f"l{l}.{nm}"))
        out.append(mk(*cfg["mlp_down"], f"l{l}.down"))
    if n_layers is None:
        out.append(mk(*cfg["head"], "lm_head"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-2-7b")
    ap.add_argument("--layers", type=int, default=None, help="decoder layers to build (default: all + lm_head)")
    ap.add_argument("--ratio", type=float, default=0.9)
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--svd_batch", type=int, default=16)
    ap.add_argument("--stable_rank", action="store_true", help="time calib_sensitivity_stable_rank (values-only sigma_max per layer) instead")
    ap.add_argument("--full_rank", action="store_true", help="factorise all min(m,n) triplets instead of the rank needed at --ratio")
    args = ap.parse_args()
    from asvd4llm_amd import _lib, ops
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    from asvd4llm_amd.parallel import svd_flops
    _lib.load(require_device=True)
    os.environ["ASVD_STRICT"] = "1"
    dev = torch.device("cuda", 0)
    t0 = time.time()
    lins = build_linears(args.model, dev, args.layers)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    if args.stable_rank:
        import types
        from asvd4llm_amd.sensitivity import calib_sensitivity_stable_rank
        holder = nn.Module()
        holder.layers = nn.ModuleList(lins)
        holder.config = types.SimpleNamespace(_name_or_path="bench/" + args.model)
        sargs = types.SimpleNamespace(scaling_method="abs_mean", alpha=args.alpha, n_calib_samples=1, calib_dataset="synthetic", svd_batch=args.svd_batch)
        os.makedirs("gpurun_out/sr", exist_ok=True); os.chdir("gpurun_out/sr")
        torch.cuda.synchronize(); t0 = time.time()
        sens = calib_sensitivity_stable_rank(holder, [{"input_ids": torch.zeros(1, 4, dtype=torch.long)}], sargs, use_cache=False)
        torch.cuda.synchronize(); t_sr = time.time() - t0
        print(json.dumps({"model": args.model, "linears": len(lins), "stable_rank_sensitivity_s": t_sr, "layers_in_dict": len(sens)}))
        return
    ranks = {l: SVDLinear.compute_rank(l, args.ratio) for l in lins}
    flops = sum(svd_flops(l.out_features, l.in_features) for l in lins)
    torch.cuda.synchronize(); t0 = time.time()
    SVDLinear.prefactorize(lins, act_aware=True, alpha=args.alpha, ranks=None if args.full_rank else ranks, max_batch=args.svd_batch)
    torch.cuda.synchronize(); t_fact = time.time() - t0
    t0 = time.time()
    sweeps = []
    for l in lins:
        for r in ([0.4, 0.5, 0.6, 0.7, 0.8, 0.9] if False else [args.ratio]):
            m = SVDLinear.from_linear(l, r, act_aware=True, alpha=args.alpha, sigma_fuse="UV")
            assert isinstance(m, SVDLinear)
        sweeps.append(l._asvd_svd_info.sweeps)
        SVDLinear.drop_factor_cache(l)
    torch.cuda.synchronize(); t_split = time.time() - t0
    out = {"model": args.model, "linears": len(lins), "ratio": args.ratio, "svd_batch": args.svd_batch, "full_rank": args.full_rank,
           "build_s": t_build, "factorize_s": t_fact, "truncate_split_s": t_split, "decompose_total_s": t_fact + t_split,
           "algorithmic_flops": flops, "achieved_TFLOPs": flops / (t_fact + t_split) / 1e12, "frac_of_157.3TF": flops / (t_fact + t_split) / 157.3e12,
           "sweeps_min_max": [min(sweeps), max(sweeps)], "max_mem_GB": torch.cuda.max_memory_allocated() / 2**30}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
