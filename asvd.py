"""asvd.py — the reference CLI (asvd.py:14-201) on the MI355X hot path.  Same flags, same pipeline order:
load model -> calibration data -> calib_input_distribution -> sensitivity sweep -> binary search + decomposition -> eval.

Additive flags only: --calib_dataset synthetic (no network in this image), --random_init <hf config name|json> to build a
shape-faithful randomly initialised model when no checkpoint is on disk, --exclude_lm_head, --dist (torchrun: one rank per
GPU, layers sharded, RCCL all-gather of sensitivities)."""
import argparse
import os

import numpy as np
import torch


def build_model(args):
    from transformers import AutoConfig, AutoModelForCausalLM, AutoTokenizer
    tokenizer = None
    if args.random_init:
        from asvd4llm_amd.model_zoo import random_init_model
        # synthetic weights are created on the GPU (seconds instead of ~2 minutes for a 7B-shaped model); ASVD_INIT_ON_CPU=1 keeps the host stream
        init_dev = None if os.environ.get("ASVD_INIT_ON_CPU") == "1" or not torch.cuda.is_available() else "cuda"
        model = random_init_model(args.model_id, dtype=torch.float16, device=init_dev)
    else:
        tokenizer = AutoTokenizer.from_pretrained(args.model_id, trust_remote_code=True)
        model = AutoModelForCausalLM.from_pretrained(args.model_id, torch_dtype=torch.float16, trust_remote_code=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    model = model.to(dev)
    return model, tokenizer


def main(args):
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)

    if args.dist:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))  # gloo: several ranks may share a GPU
        dist.init_process_group(args.dist_backend)  # "nccl" = RCCL over xGMI; "gloo": CPU rendezvous (several ranks on one GPU, tests)

    from asvd4llm_amd.act_aware_utils import calib_fisher_info, calib_input_distribution
    from asvd4llm_amd.binary_search import binary_search_truncation_rank
    from asvd4llm_amd.datautils import get_calib_data
    from asvd4llm_amd.evaluate_utils import evaluate_model
    from asvd4llm_amd.sensitivity import calib_sensitivity_ppl, calib_sensitivity_stable_rank

    model, tokenizer = build_model(args)
    if args.exclude_lm_head:
        from asvd4llm_amd.model_zoo import hide_lm_head
        hide_lm_head(model)

    if not args.raw_model:
        calib_loader = get_calib_data(args.calib_dataset, tokenizer, args.model_id, args.n_calib_samples, seed=args.seed, use_bos=args.use_bos,
                                      vocab_size=model.config.vocab_size)
        if "fisher" in args.scaling_method:
            calib_fisher_info(model, calib_loader, args.use_cache)
        if "abs" in args.scaling_method:
            calib_input_distribution(model, calib_loader, args.scaling_method, args.use_cache, shard_samples=args.shard_calib)
        if args.sensitivity_metric == "ppl":
            sensitivity = calib_sensitivity_ppl(model, calib_loader, args, args.use_cache)
        elif args.sensitivity_metric == "stable_rank":
            sensitivity = calib_sensitivity_stable_rank(model, calib_loader, args, args.use_cache)

        binary_search_truncation_rank(model, sensitivity, calib_loader, args)

        if args.weight_quant != "none":
            raise NotImplementedError("weight quantization (rtn/awq) is outside the hot-path scope of this build (SURVEY.md §2)")

        if args.save_repo and (not args.dist or int(os.environ.get("RANK", "0")) == 0):
            # exported-repo layout of huggingface_repos/build_asvd_repo.py:58-92 (truncation_ranks + ALinear/BLinear keys)
            from asvd4llm_amd.export import save_asvd_repo
            ranks = save_asvd_repo(model, args.save_repo, tokenizer if hasattr(tokenizer, "save_pretrained") else None)
            print(f"saved ASVD repo with {len(ranks)} factorised layers to {args.save_repo}")

    if args.dist and int(os.environ.get("RANK", "0")) != 0 and args.gather_factors != "all":
        # only rank 0 holds the complete compressed model (factors gathered point-to-point): it alone evaluates and reports
        import torch.distributed as dist
        dist.destroy_process_group()  # no collective follows the factor exchange
        return
    eval_ids = torch.cat([_["input_ids"] for _ in calib_loader], 0) if (not args.raw_model and args.calib_dataset == "synthetic") else None
    result = evaluate_model(model, tokenizer, args.model_id, "mmlu" if args.eval_mmlu else args.eval_tasks, eval_ppl=args.eval_ppl, limit=-1,
                            use_bos=args.use_bos, eval_ids=eval_ids)
    print(result)
    if not args.dist or int(os.environ.get("RANK", "0")) == 0:
        if not os.path.exists("output"):
            os.makedirs("output")
        with open("output/result.txt", "a+") as f:
            f.write(f"{args}\n")
            f.write(f"{result}\n")
    if args.dist:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--model_id", type=str, default="facebook/opt-1.3b", help="Pretrained model ID")
    parser.add_argument("--ppl_target", type=float, default=-1, help="target ppl")
    parser.add_argument("--param_ratio_target", type=float, default=-1, help="target param ratio")
    parser.add_argument("--act_aware", action="store_true", help="use act aware svd (ASVD)")
    parser.add_argument("--alpha", type=float, default=0.5, help="hyper-parameter alpha for ASVD")
    parser.add_argument("--n_calib_samples", type=int, default=32, help="number of samples used for calibration")
    parser.add_argument("--calib_dataset", type=str, default="wikitext2",
                        choices=["wikitext2", "c4", "ptb", "alpaca", "selfgen", "synthetic"], help="calibration dataset")
    parser.add_argument("--scaling_method", type=str, default="abs_mean", choices=["abs_mean", "abs_max", "fisher", "fisher_abs_mean"],
                        help="scaling method")
    parser.add_argument("--sensitivity_metric", type=str, default="ppl", choices=["ppl", "stable_rank"], help="search metric")
    parser.add_argument("--use_cache", action="store_true", help="use cached calibration results")
    parser.add_argument("--weight_quant", type=str, default="none", choices=["none", "rtn_int8", "rtn_int6", "awq_int8", "awq_int4"],
                        help="weight quantization method")
    parser.add_argument("--eval_mmlu", action="store_true", help="evaluate mmlu")
    parser.add_argument("--eval_ppl", default="wikitext2,ptb", type=str)
    parser.add_argument("--eval_tasks", type=str, default="")
    parser.add_argument("--sigma_fuse", type=str, default="UV", help="sigma fuse method", choices=["U", "V", "UV"])
    parser.add_argument("--seed", type=int, default=233, help="random seed, which can significantly affect the calibration results")
    parser.add_argument("--compress_kv_cache", action="store_true", help="compress kv cache by asvd for k_proj and v_proj")
    parser.add_argument("--kv_cache_ratio_target", type=float, default=-1, help="kv cache ratio")
    parser.add_argument("--rank_align", type=int, default=1, help="align rank in SVD")
    parser.add_argument("--raw_model", action="store_true", help="use the raw model without ASVD")
    parser.add_argument("--use_bos", action="store_true", help="use bos token in calibration")
    # ---- additive, build-only flags ----
    parser.add_argument("--random_init", action="store_true", help="shape-faithful random-init model named by --model_id (no checkpoints offline)")
    parser.add_argument("--exclude_lm_head", action="store_true", help="do not hook/sweep/compress lm_head (the reference includes it)")
    parser.add_argument("--save_repo", type=str, default="", help="write the compressed model in the exported HF-repo layout of the reference (truncation_ranks in config.json)")
    parser.add_argument("--no_fused_sweep", dest="fused_sweep", action="store_false",
                        help="evaluate every (layer, ratio) with full model forwards as the reference does (default: prefix-cached evaluator, same values)")
    parser.add_argument("--no_fused_ratios", dest="fused_ratios", action="store_false",
                        help="evaluate every candidate ratio of a layer in its own suffix pass (default: all ratios as one batched pass; "
                             "same search trace, perplexities equal to ~1e-4 relative in fp16)")
    parser.add_argument("--sweep_samples_per_pass", type=int, default=1,
                        help="calibration samples per batched suffix pass of the sensitivity sweep (ratios x samples rows per GEMM).  1 (default) keeps the "
                             "perplexities and the search trace of rounds 2-5 bit for bit; 4 shortens the Llama-2-7B-shaped sweep by 6.5 %% (712 vs 762 s, "
                             "profiles/r6_e2e_llama2_7b_ncalib32_*.json) with the same per-sample arithmetic on 4x taller GEMMs: perplexities equal to "
                             "<= 5e-4 relative in fp16, which can reorder near-tied layers in the search")
    parser.add_argument("--gather_factors", type=str, default="rank0", choices=["rank0", "all", "none"],
                        help="--dist: after the sharded decomposition send every layer's A/B factors to rank 0 (point-to-point), to all ranks "
                             "(broadcast), or nowhere")
    parser.add_argument("--dist", action="store_true", help="torchrun launch: one rank per GPU, layers sharded, RCCL all-gather of sensitivities")
    parser.add_argument("--dist_backend", type=str, default="nccl", choices=["nccl", "gloo"],
                        help="--dist: process-group backend; nccl is RCCL over xGMI (one rank per GPU), gloo lets several ranks share one GPU")
    parser.add_argument("--shard_calib", action="store_true",
                        help="--dist: every rank runs the calibration hook pass over its own samples and the [C] accumulators are all-reduced "
                             "(default: replicated pass, identical statistics without a collective)")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
