"""bench.py — headline benchmark of the ASVD hot path on MI355X.

metric (BASELINE.json): weight-matrix SVDs/sec on 4096x4096 fp32.  A "step" is one pass of the hot path over one batch of
synthetic Linears already resident in HBM: scale vector from abs_mean statistics -> W*diag(s) (fused in the pack kernel) ->
exact economy SVD (hand-written block Jacobi) -> rank-512 truncation / un-scale / sigma-fuse / fp16 cast (BASELINE configs[1]).
Nothing is cached between steps.  N>1: one process per GPU (torchrun), each rank factorises its own batch — the path
shards over independent matrices with no data-path collective (weak scaling); only barrier + max-reduce of the time.

Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth(size_m, size_n, seed, n_calib=32):
    """SURVEY.md §8d synthetic 'LLM-like' Linear: W~N(0,0.02^2) with 0.5% outlier columns x20, abs-mean statistics
    n_calib*|N(0,1)| with 1% channels x30 (fp16)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(size_m, size_n, generator=g) * 0.02
    k = max(1, int(0.005 * size_n))
    W[:, torch.randperm(size_n, generator=g)[:k]] *= 20
    scal = n_calib * torch.randn(size_n, generator=g).abs()
    k = max(1, int(0.01 * size_n))
    scal[torch.randperm(size_n, generator=g)[:k]] *= 30
    return W.float(), scal.to(torch.float16)


MODEL_SHAPES = {  # (out, in) of the Linears of a decoder layer + lm_head, and the layer count (model configs[2]-[4])
    "llama-2-7b": dict(layers=32, attn=(4096, 4096), up=(11008, 4096), down=(4096, 11008), head=(32000, 4096)),
    "llama-2-13b": dict(layers=40, attn=(5120, 5120), up=(13824, 5120), down=(5120, 13824), head=(32000, 5120)),
    "llama-7b-2layers": dict(layers=2, attn=(4096, 4096), up=(11008, 4096), down=(4096, 11008), head=(32000, 4096)),   # tests: the 7B shapes, 15 Linears
    "tiny": dict(layers=3, attn=(64, 64), up=(176, 64), down=(64, 176), head=(320, 64)),
}


def model_linears(name):
    """[(full_name, out, in)] in the order the reference's sweep visits a Llama module tree (lm_head, then the layers from the last to the
    first, mlp before self_attn: sensitivity.py:19-33 as pinned by tests/golden/linear_order_hf.json)"""
    c = MODEL_SHAPES[name]
    out = [("lm_head",) + c["head"]]
    for l in range(c["layers"] - 1, -1, -1):
        out += [(f"model.layers.{l}.mlp.gate_proj",) + c["up"], (f"model.layers.{l}.mlp.up_proj",) + c["up"], (f"model.layers.{l}.mlp.down_proj",) + c["down"]]
        out += [(f"model.layers.{l}.self_attn.{p}_proj",) + c["attn"] for p in ("q", "k", "v", "o")]
    return out


def model_sweep_costs(name, n_ratios=6, n_samples=32, seqlen=2048):
    """predicted seconds of every layer's share of the ppl sweep (parallel.sweep_layer_costs) for the Linears of model_linears(name)"""
    from asvd4llm_amd import parallel
    c = MODEL_SHAPES[name]
    block_params = 4 * c["attn"][0] * c["attn"][1] + 2 * c["up"][0] * c["up"][1] + c["down"][0] * c["down"][1]
    lay = [(o, i, "after" if n == "lm_head" else int(n.split(".")[2])) for n, o, i in model_linears(name)]
    return parallel.sweep_layer_costs(lay, c["layers"], block_params, c["head"][0] * c["head"][1], n_ratios, n_samples, seqlen,
                                      4.0 * c["attn"][1] * seqlen / 2.0)


def predicted_sweep_balance(world=8, models=("llama-2-7b", "llama-2-13b")):
    """BASELINE configs[3] / [4] are 99 % sweep: the LPT shard of calib_sensitivity_ppl is balanced on the predicted sweep seconds per layer
    (suffix forwards + factorisation), the final decomposition on SVD flops.  Host arithmetic only — reported so that the first 8-GPU run
    does not have to discover a straggler."""
    from asvd4llm_amd import parallel
    out = {}
    for name in models:
        costs = model_sweep_costs(name)
        flops = [parallel.svd_flops(o, i) for _, o, i in model_linears(name)]
        own = parallel.lpt_assign(costs, world)
        own_f = parallel.lpt_assign(flops, world)
        out[name] = {"ranks": world, "n_calib": 32, "seqlen": 2048, "ratios": 6, "predicted_sweep_s_one_gpu": sum(costs),
                     "predicted_sweep_s_per_rank_max": max(sum(c for c, o in zip(costs, own) if o == r) for r in range(world)),
                     "sweep_s_max_over_mean": parallel.load_balance(costs, own, world),
                     "sweep_s_max_over_mean_if_sharded_on_svd_flops": parallel.load_balance(costs, own_f, world),
                     "decompose_flops_max_over_mean": parallel.load_balance(flops, own_f, world)}
    return out


class _Slot:
    """stand-in for the father module of a layer in the factor-exchange leg (parallel.exchange_factors only needs getattr / setattr)"""


class _RawShape:
    """what exchange_factors reads of a raw nn.Linear it does not own: shape, dtype and device of the weight"""

    def __init__(self, out_features, in_features, dev, dtype):
        import torch
        self.out_features, self.in_features = out_features, in_features
        self.weight = torch.empty(0, dtype=dtype, device=dev)


def sharded_model_leg(name, rank, world, dev, ratio=0.9, dry=False, samples=None):
    """BASELINE configs[3]: the Linears of a Llama-2-7B-shaped model LPT-sharded over the ranks, every rank decomposes its own layers with the HIP
    path, ONE all-gather of the per-layer sensitivities on the process group (RCCL when the bench runs on GPUs), then the replicated search.
    Untimed extra of `bench.py --gpus N` (not part of `value`); returns the record rank 0 prints under "sharded_model" — at ONE GPU it is the
    whole-model decomposition of configs[2], printed under "full_model": the second component of BASELINE.json's metric ("full-model ASVD
    wall-clock"; the stage the reference itself times, binary_search.py:111-131).  dry: no kernels (gloo plumbing test): the sensitivities are
    made up, everything else is the real code path.  samples: a list that receives, for the first layer of every distinct shape, what the
    CPU-oracle parity check of the caller needs (weight, statistics, spectrum and emitted factors, all still on the device)."""
    import hashlib
    import zlib
    import torch
    import torch.distributed as dist
    from asvd4llm_amd import parallel
    from asvd4llm_amd.binary_search import _CutPlans, _bisect_cut, _plan_params
    layers = model_linears(name)
    names = [n for n, _, _ in layers]
    costs = [parallel.svd_flops(o, i) for _, o, i in layers]
    owner = parallel.lpt_assign(costs, world)
    ratios = [0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
    load = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(world)]
    mine = [(n, o, i) for (n, o, i), ow in zip(layers, owner) if ow == rank]
    local, t_dec, sweeps, err, mods = {}, 0.0, [], None, {}
    if dry:
        for n, o, i in mine:
            local[n] = {r: 5.0 + (zlib.crc32(f"{n}:{r}".encode()) % 1000) / 1000.0 for r in ratios}
    else:
        # a failure of this rank's share must not leave the other ranks waiting in the collectives below (nor cost the bench line): it is
        # recorded, the rank contributes NaNs, every rank learns about it from the reduction at the end
        at_start = False
        try:
            import torch.nn as nn
            from asvd4llm_amd.modules.svd_linear import SVDLinear
            lins = []
            for idx, (n, o, i) in enumerate(mine):
                g = torch.Generator(device=dev).manual_seed(233 + 7919 * rank + idx)
                lin = nn.Linear(i, o, bias=False, device="meta")
                lin.weight = nn.Parameter((torch.randn(o, i, generator=g, device=dev) * 0.02).half(), requires_grad=False)
                scal = 32 * torch.randn(i, generator=g, device=dev).abs()
                scal[torch.randperm(i, generator=g, device=dev)[:max(1, i // 100)]] *= 30
                lin.scaling_diag_matrix = scal.half()
                lins.append(lin)
            ranks = {l: SVDLinear.compute_rank(l, ratio) for l in lins}
            torch.cuda.synchronize()
            at_start = True
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            SVDLinear.prefactorize(lins, act_aware=True, alpha=0.5, ranks=ranks, max_batch=32)
            seen_shapes = set()
            for (n, o, i), l in zip(mine, lins):
                mod = SVDLinear.from_linear(l, ratio, act_aware=True, alpha=0.5, sigma_fuse="UV")
                assert isinstance(mod, SVDLinear) and l._asvd_svd_info.status == 0, n
                sweeps.append(l._asvd_svd_info.sweeps)
                S = l._asvd_factor_cache[1][1].float()
                mods[n] = mod
                if samples is not None and (o, i) not in seen_shapes:
                    seen_shapes.add((o, i))
                    samples.append({"name": n, "shape": [o, i], "rank": ranks[l], "W": l.weight.data, "stat": l.scaling_diag_matrix, "S": S.clone(),
                                    "A": mod.ALinear.weight.data, "B": mod.BLinear.weight.data, "sweeps": l._asvd_svd_info.sweeps})
                # a sensitivity made of the layer's own spectrum (the real sweep would put calibration perplexities here): what matters is that real
                # per-layer numbers computed on the owning rank cross the collective
                local[n] = {r: float(1.0 + S[min(SVDLinear.compute_rank(l, r), S.numel()) - 1] / S[0]) for r in ratios}
                SVDLinear.drop_factor_cache(l)
            torch.cuda.synchronize()
            t_dec = time.perf_counter() - t0
        except Exception as e:   # noqa: BLE001 — reported in the record
            err = f"{type(e).__name__}: {e}"[:300]
            local = {n: {r: float("nan") for r in ratios} for n, _, _ in mine}
            sweeps = []
            if world > 1 and not at_start:   # failed while building its layers: still meet the others at the start line
                dist.barrier()
    # ---- the one exchange step of the path: all-gather of the sensitivities ----
    if world > 1:
        dist.barrier()
    if not dry:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    full = parallel.allgather_sensitivities(local, names, ratios, owner)
    if not dry:
        torch.cuda.synchronize()
    t_ag = time.perf_counter() - t0
    t0 = time.perf_counter()
    full2 = parallel.allgather_sensitivities(local, names, ratios, owner)  # second call: without communicator set-up
    if not dry:
        torch.cuda.synchronize()
    t_ag2 = time.perf_counter() - t0
    assert list(full.keys()) == names and list(full2.keys()) == names
    broken = any(v != v for d in full.values() for v in d.values())   # a rank contributed NaNs (its share failed)
    assert broken or full2 == full
    # ---- replicated search on the complete dict: every rank must arrive at the same plan ----
    weights = {n: o * i for n, o, i in layers}
    plan, digest = {}, 0
    if not broken:
        plans = _CutPlans(full, lambda r: r < 1, 1)
        cut = _bisect_cut(plans.size, lambda lo, mid, hi: (lambda ct: ct[0] / ct[1] > ratio)(_plan_params(plans.plan(mid), weights)))
        plan = plans.plan(cut)
        digest = int(hashlib.sha256(json.dumps(sorted(plan.items())).encode()).hexdigest()[:15], 16)
    # ---- the only multi-GB transfer of the path: the owners' factors to rank 0 (parallel.exchange_factors, mode "rank0": every receive and every
    # send posted at once, so all links into rank 0 are driven together) ----
    t_gather, gathered, gather_bytes = 0.0, 0, 0
    if world > 1 and not dry and not broken and not err:
        try:
            items, own_by_name = [], {}
            for (n, o, i), ow in zip(layers, owner):
                slot = _Slot()
                if ow == rank:
                    slot.m = mods[n]
                items.append((n, slot, "m", _RawShape(o, i, dev, torch.float16)))
                own_by_name[n] = ow
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gathered = parallel.exchange_factors(items, own_by_name, mode="rank0")
            torch.cuda.synchronize()
            dist.barrier()
            t_gather = time.perf_counter() - t0
            if rank == 0:
                gather_bytes = sum(2 * (it[1].m.ALinear.weight.numel() + it[1].m.BLinear.weight.numel()) for it in items if own_by_name[it[0]] != 0)
        except Exception as e:   # noqa: BLE001 — reported, the bench line survives
            err = f"factor gather: {type(e).__name__}: {e}"[:300]
    mods.clear()
    comm_dev = torch.device("cpu")   # the bookkeeping reduction of this record runs on the default (gloo, host) group
    red = torch.tensor([float(digest), -float(digest), t_dec, t_ag, t_ag2, 1.0 if err else 0.0, t_gather, float(gather_bytes)], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    red = red.cpu()
    if broken or float(red[5]) > 0:
        return {"model": name, "error": err or "another rank's share failed (see its stderr)", "ranks_failed": bool(float(red[5]) > 0),
                "collective_world_size": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1}
    comp, total = _plan_params(plan, weights)
    return {"model": name, "config": "BASELINE %s: every Linear of the model, ratio %.2f, alpha 0.5, synthetic weights / statistics" % ("configs[3] (layers sharded over the ranks)" if world > 1 else "configs[2] (all Linears on one GPU)", ratio),
            "linears": len(layers), "collective_world_size": dist.get_world_size() if (world > 1 or (dist.is_available() and dist.is_initialized())) else 1,
            "collective_backend": (parallel.backend() if (dist.is_available() and dist.is_initialized()) else "none (single process)") + (" = RCCL" if (dist.is_available() and dist.is_initialized() and parallel.backend() == "nccl") else ""),
            "layers_per_rank": [sum(1 for o in owner if o == r) for r in range(world)],
            "load_flops_max_over_mean": max(load) / (sum(load) / world), "decompose_s_max_over_ranks": float(red[2]),
            "svd_flops_total": sum(costs), "achieved_TFLOPs_whole_job": (sum(costs) / float(red[2]) / 1e12) if float(red[2]) > 0 else None,
            "allgather_first_call_ms": 1e3 * float(red[3]), "allgather_ms": 1e3 * float(red[4]),
            "allgather_payload_bytes_per_rank": 2 * 8 * max(1, max(sum(1 for o in owner if o == r) for r in range(world)) * len(ratios)),
            "plan_identical_on_all_ranks": bool(float(red[0]) == -float(red[1])), "plan_param_ratio": comp / total,
            "layers_factorised_by_plan": sum(1 for v in plan.values() if v < 1), "sweeps_min_max_rank0": [min(sweeps), max(sweeps)] if sweeps else None,
            "decompose_s": float(red[2]), "decompose_stage": "scale + factorise + truncate / split of every Linear (what the reference times as `decompose time`, binary_search.py:111-131)",
            "gather_factors_s": float(red[6]) if world > 1 else None, "gather_factors_bytes_into_rank0": int(red[7]) if world > 1 else None,
            "gather_factors_GBps_into_rank0": (float(red[7]) / float(red[6]) / 1e9) if (world > 1 and float(red[6]) > 0) else None,
            "predicted_sweep_balance_8_ranks": predicted_sweep_balance(8)}


def full_model_parity(samples, model_name, dev, threads, decompose_s=None):
    """Per-shape parity of the model leg against the CPU oracle, and the CPU reference the full-model wall-clock divides: the oracle pipeline
    (scale + torch.linalg.svd + truncate / split) timed ONCE per distinct shape at `threads` host threads, times the number of Linears of that
    shape ("per-shape x count": the whole model on the host would be ~13 minutes).  Checker code: the only place of this file besides the
    cpu_baseline leg that touches oracle/ (bench.py's CPU leg and tests/test_gpu_full_model.py call it)."""
    import torch
    from oracle import asvd_oracle as O
    torch.set_num_threads(threads)
    counts = {}
    for _, o_, i_ in model_linears(model_name):
        counts[(o_, i_)] = counts.get((o_, i_), 0) + 1
    per_shape, cpu_total = [], 0.0
    for smp in samples:
        o_, i_ = smp["shape"]
        Wc, stc = smp["W"].cpu(), smp["stat"].cpu()
        sc_ = O.make_scale(stc, 0.5)
        t1 = time.perf_counter()
        ws_ = O.scaled_weight(Wc, sc_)
        Uo, So, Vo = O.exact_svd(ws_)
        Ao, Bo, _ = O.truncate_split(Uo, So, Vo, sc_, smp["rank"], "UV", torch.float16)
        t_shape = time.perf_counter() - t1
        cpu_total += t_shape * counts[(o_, i_)]
        # reconstruction parity in fp64 ON THE DEVICE (checker arithmetic: torch matmul; the factors under test came from the HIP path)
        live = O.live_channels(sc_).to(dev)
        sd = sc_.double().to(dev).flatten()
        D = smp["A"].double() @ smp["B"].double() - Ao.to(dev).double() @ Bo.to(dev).double()
        Wd = smp["W"].double()
        rec_live = float((D[:, live].norm() / Wd.norm()).item())
        rec_scaled = float(((D * sd).norm() / (Wd * sd).norm()).item())
        del D, Wd
        per_shape.append({"shape": [o_, i_], "layer": smp["name"], "count_in_model": counts[(o_, i_)], "rank": smp["rank"], "sweeps": smp["sweeps"],
                          "sigma_rel_err_top_r": O.sigma_rel_err(smp["S"].cpu(), So, smp["rank"]), "recon_fro_err_vs_oracle": rec_live,
                          "recon_fro_err_scaled_norm": rec_scaled, "cpu_oracle_seconds": t_shape})
    return {"parity_per_shape": per_shape,
            "parity_ok": all(p_["sigma_rel_err_top_r"] <= 1e-4 and p_["recon_fro_err_vs_oracle"] <= 1e-3 for p_ in per_shape),
            "parity_tolerance": {"sigma": 1e-4, "recon": 1e-3, "note": "north_star's bars; the factors compared are the emitted fp16 ones on both sides"},
            "cpu_reference": {"seconds_whole_model_per_shape_x_count": cpu_total, "threads": threads, "kind": "port",
                              "method": "oracle scale + torch.linalg.svd + truncate/split timed once per distinct shape on this host, x the number of Linears of that shape"},
            "speedup_vs_cpu_reference": (cpu_total / decompose_s) if decompose_s else None}


def dry_run(args, rank, world):
    """CPU/gloo exercise of everything around the kernels: rank binding, rendezvous, barrier-bracketed timing, MAX over ranks, one
    JSON line on rank 0.  No SVD runs (the product path has no CPU fallback), so value is null."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(rank)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sharded = None
    if args.sharded_model != "none":
        sharded = sharded_model_leg("tiny" if args.sharded_model == "auto" else args.sharded_model, rank, world, torch.device("cpu"), dry=True)
    if rank == 0:
        assert world == 1 or int(t[1].item()) == world - 1
        print(json.dumps({"metric": "weight-matrix SVDs/sec (4096x4096 fp32)", "value": None, "unit": "SVD/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "dry_run": True, "sharded_model": sharded,
                          "config": {"workload": "dry run: launch plumbing only", "batch_per_gpu": args.batch, "parallelism": f"independent matrices x{world}"}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="matrices factorised concurrently per step and GPU (same-shape problems share every launch; 32 = the q/k/v/o Linears of eight layers, the pipeline's default svd_batch)")
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--rank", type=int, default=512)
    ap.add_argument("--prewarm_s", type=float, default=5.0, help="seconds of untimed identical work before the warm-up steps (0 disables)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_latency", action="store_true", help="skip the batch-1 latency leg (profiling runs: keeps per-kernel averages to the batch workload)")
    ap.add_argument("--cpu_reps", type=int, default=3, help="repetitions of the CPU oracle pipeline at the best thread count (>= 3; median reported)")
    ap.add_argument("--cpu_budget_s", type=float, default=90.0, help="soft bound on the CPU-baseline leg (warm-up + thread sweep + repetitions)")
    ap.add_argument("--sharded_model", default="auto", help="untimed extra: decomposition of a whole model's Linears, LPT-sharded over the ranks, + the sensitivity "
                    "all-gather and the factor gather on the process group (BASELINE configs[2] at one GPU -> \"full_model\", configs[3] at N -> \"sharded_model\"); "
                    "auto = llama-2-7b; or llama-2-7b / llama-2-13b / tiny / none")
    ap.add_argument("--sharded_timeout_s", type=float, default=420.0, help="give the untimed whole-model leg (N > 1: incl. creating the RCCL group) this long, then print the bench line without it")
    ap.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"], help="N > 1: backend of the DEVICE group of the sharded-model leg — nccl = RCCL over xGMI (the "
                    "product; falls back to gloo, and says so, when the group cannot be created); gloo = CI stand-in, together with --same_gpu it runs the N ranks of "
                    "the whole bench — sharded model, all-gather, factor gather — with the real kernels on ONE GPU.  The timed region's barriers and the MAX-reduce "
                    "always run on a gloo host group")
    ap.add_argument("--same_gpu", action="store_true", help="bind every rank to cuda:0 (with --dist_backend gloo)")
    ap.add_argument("--dry_run", action="store_true", help="launch plumbing only (CPU, gloo): spawn/bind ranks, barrier, max-reduce, JSON line; no kernels, value = null")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a torchrun environment: launch the N ranks ourselves (one process per GPU, RCCL rendezvous on
    # 127.0.0.1) and let rank 0 of the children print the JSON line.  Under the driver's own `python -m torch.distributed.run ...`
    # WORLD_SIZE is already set and this branch is skipped.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    from asvd4llm_amd import _lib, ops
    from asvd4llm_amd.parallel import svd_flops

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world} (the ranks that actually run)", file=sys.stderr)
    if args.dry_run:
        return dry_run(args, rank, world)
    gpu_index = 0 if (world == 1 or args.same_gpu) else local_rank
    if world > 1:
        torch.cuda.set_device(gpu_index)
        # The DEFAULT group is gloo, always: rendezvous, the barriers around the timed region, the MAX-reduce of the time and the per-rank
        # rates are host-side and tiny — the weak-scaling number needs no device collective, so a first RCCL problem on a node nobody has
        # touched cannot cost it.  RCCL (--dist_backend nccl) is a SECOND group, created inside the watchdogged sharded-model leg below, where
        # the path's one real exchange step (the sensitivity all-gather) and the factor gather run on it.
        dist.init_process_group("gloo")
    dev = torch.device("cuda", gpu_index)
    comm_dev = torch.device("cpu")
    _lib.load(require_device=True)  # fails loudly without the HIP library / a gfx950 device

    B, m, n, r = args.batch, args.m, args.n, args.rank
    mats, stats = [], []
    for b in range(B):
        W, scal = synth(m, n, seed=233 + 1000 * rank + b)
        mats.append(W.to(dev))
        stats.append(scal.to(dev))
    torch.cuda.synchronize()

    def step():
        scales = ops.make_scale_batched(stats, alpha=0.5)
        U, S, V, infos = ops.svd_batched(mats, scales)
        As, Bs, flags = ops.truncate_split_batched(U, S, V, scales, r, "UV", torch.float16)
        outs = [(As[b], Bs[b], flags[b]) for b in range(B)]
        return U, S, V, scales, outs, infos

    # A fresh process on a fresh box runs its first ~2 s below steady state (clock ramp, first-touch of the multi-GB workspace):
    # measured 18.8-21.1 vs 23.3 SVD/s for the same binary.  Spin the same workload untimed until the GPU has been busy for
    # PREWARM_S seconds before the W warm-up steps the contract asks for; reported as "prewarm_steps".
    prewarm_steps = 0
    t_pw = time.perf_counter()
    # The untimed steps hold their result while the next one runs, exactly as the timed loop does: the caching allocator then owns
    # both sets of output buffers before the clock starts (measured: otherwise the SECOND timed step pays ~300 ms of hipMalloc for
    # the 3 GB of U/S/V it cannot reuse yet — 413, 751, 413, 414 ms on a fresh box).
    res = None
    while time.perf_counter() - t_pw < args.prewarm_s:
        res = step()
        torch.cuda.synchronize()
        prewarm_steps += 1
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        res = step()
        step_marks.append(time.perf_counter())  # host time at which the step's last launch was queued (the SVD call itself ends synchronised)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_local = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the outputs every parity check below looks at are those of the LAST TIMED step (the split call when the batch qualifies), not of the profiled one
    U, S, V, scales, outs, timed_infos = res
    res = None
    # ---- per-kernel-class durations with HIP events on the launch stream (one extra, untimed, profiled step) ----
    ops.svd_profile(True)
    infos = step()[5]
    torch.cuda.synchronize()
    prof = ops.svd_profile()
    ops.svd_profile(False)
    assert all(i.status == 0 for i in infos) and all(i.status == 0 for i in timed_infos), [i.status for i in infos]  # every SVD of the batch converged
    # ---- the same, in the configuration the TIMED steps ran in: a profiled step that keeps the split over the two chip halves (both halves time
    # their own launches with HIP events on their own CU-masked stream; asvd_svd_set_profiling mode 2) ----
    split_prof, split_wall_ms = None, None
    if any(i.split for i in timed_infos):
        ops.svd_profile(True, keep_split=True)
        torch.cuda.synchronize()
        t_sp = time.perf_counter()
        sp_res = step()
        torch.cuda.synchronize()
        split_wall_ms = 1e3 * (time.perf_counter() - t_sp)
        split_tot = ops.svd_profile()
        split_prof = ops.svd_split_profile()
        ops.svd_profile(False)
        if split_prof is not None:
            split_prof["pairs"] = split_tot["pairs"]
        del sp_res

    # per-rank rates (N > 1): every rank's own SVDs/s over its own wall clock
    per_rank = None
    if world > 1:
        mine = torch.tensor([B * args.steps / dt_local], dtype=torch.float64, device=comm_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]

    out = None
    if rank == 0:
        total_svds = B * args.steps * world
        value = total_svds / dt
        f_svd = svd_flops(m, n)
        pairs_cnt = prof.pop("pairs")
        sweep_ms, sweep_rot = prof.pop("sweep_ms"), prof.pop("sweep_rotated")
        classes = {k: {"ms_per_step": v["ms"], "launches": v["launches"], "avg_us": (1e3 * v["ms"] / v["launches"]) if v["launches"] else 0.0}
                   for k, v in prof.items()}
        # Jacobi runs on the square Cholesky factor (cols_pad rows) when the problem has >= 128 columns, else on the matrix itself
        rows_pad = ((max(m, n) + 31) // 32) * 32
        cols_pad = ((min(m, n) + 63) // 64) * 64
        rows_j = cols_pad if min(m, n) >= 128 else rows_pad
        # ALGORITHMIC HBM bytes of the streaming kernels, from the library's own counters of this step:
        #   supdate (two-level update pass)  reads + writes the 128 columns of every super-pair it updates:  2 * rows * 128 * 4 B each
        #   sgram   (two-level Gram pass)    reads the 128 columns of every super-pair of the step:             rows * 128 * 4 B each
        #   update1 / gram1 (single-level kernels: internal step, sparse tail sweeps): 64 columns per rotated / visited 32-panel pair
        nb_ = cols_pad // 32
        two_level = classes["supdate"]["launches"] + classes["supgram"]["launches"] > 0
        alg = {}
        if two_level:
            ns_ = nb_ // 2
            upd_launches = classes["supdate"]["launches"] + classes["supgram"]["launches"]
            wr = 1.0 * rows_j * 128 * 4 * pairs_cnt["super_updates"] / upd_launches  # bytes written per update launch (rotated super-pairs only)
            all_x = 1.0 * rows_j * 128 * 4 * (B * (ns_ // 2))  # one read of every panel of every problem of the batch
            alg["supdate"] = 2.0 * wr             # plain update pass: reads what it writes
            alg["sgram"] = all_x                  # stand-alone Gram pass (first super-step of a sweep only, when the fused kernel is on)
            alg["supgram"] = all_x + wr           # fused update + next-step Gram: ONE read of everything, writes the rotated pairs
        dom = max((k for k in ("supgram", "supdate", "sgram", "update1", "gram1") if classes[k]["launches"]), key=lambda k: classes[k]["ms_per_step"])
        dom_all = max(("sgram", "evd", "supdate", "supgram", "gram1", "update1", "snapshot"), key=lambda k: classes[k]["ms_per_step"])
        # HBM traffic of the dominant kernel comes from PMC counters, which need their own rocprofv3 passes (tools/prof_final.sh): the
        # stored figure is only quoted when it was collected from THIS build of the library (sha256 of libasvd_hip.so recorded with it)
        # on this workload — a kernel change since then prints null instead of a stale "measured" number
        traffic, traffic_src, mfma_busy, pmc_clock, pmc_per_svd, lib_sha = None, None, None, None, None, None
        try:
            import hashlib
            lib_sha = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if dom in pmc.get("kernels", {}) and (m, n, B) == (4096, 4096, pmc.get("batch")):
                if pmc.get("lib_sha256") == lib_sha:
                    traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
                    mfma_busy = pmc["kernels"][dom].get("mfma_busy_frac")
                    pmc_clock = pmc["kernels"][dom].get("shader_clock_GHz_under_pmc")
                    pmc_per_svd = pmc.get("hbm_bytes_per_svd_all_kernels")
                    traffic_src = "stored: " + pmc.get("source", "profiles/pmc_traffic.json") + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / MFMA-busy / GRBM passes of this command, same libasvd_hip.so)"
                else:
                    traffic_src = "not quoted: profiles/pmc_traffic.json was collected from a different build of libasvd_hip.so (re-run tools/reproduce_evidence.sh prof)"
        except Exception:
            pass
        if dom in alg:
            ach = alg[dom] / (classes[dom]["avg_us"] * 1e-6) / 1e9
            roofline = {"bound": "hbm", "kernel": {"supdate": "supdate_split_kernel", "sgram": "sgram6_kernel", "supgram": "supgram_kernel"}[dom], "achieved": ach, "peak": 8000.0,
                        "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                        "frac_of_measured_copy_rate_6290GBps": ach / 6290.0, "matrix_pipe_busy_frac": mfma_busy, "shader_clock_GHz_under_pmc": pmc_clock,
                        "algorithmic_bytes_per_launch": alg[dom], "avg_launch_us": classes[dom]["avg_us"],
                        "bound_evidence": "HBM path first, shader clock second: measured traffic is 1.02x the algorithmic bytes, the fp16 matrix pipe is a quarter busy (33 MFMAs "
                                          "per wave and tile), removing the panel stores is the only ablation that shortens a launch (DESIGN.md 3.10).  Clock: in this bench, "
                                          "with lighter kernels between its launches, the kernel runs at 2.2-2.3 GHz (877-881 us; the PMC passes read 2.20-2.31 GHz); launched back "
                                          "to back it holds the socket at its 1400 W cap at 1.87-1.95 GHz and takes 957-968 us (profiles/r4_supgram_micro.jsonl, r4_power.jsonl): "
                                          "20 % of clock are worth 9 % of time.  Round 5 (DESIGN.md 3.11): on CU-masked streams the kernel delivers its full-chip throughput with "
                                          "~190 of the 256 CUs and 72 % of it with 128 — it saturates the HBM path, not the CUs",
                        "note": "dominant streaming kernel by total time; bytes from the library's own counters of the profiled step, duration = HIP "
                                "events around every launch on the launch stream, averaged over ALL its launches (one stream: nothing else runs beside it)"}
        else:
            roofline = {"bound": "hbm", "kernel": dom, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": traffic,
                        "avg_launch_us": classes[dom]["avg_us"]}
        achieved = f_svd * (value / world) / 1e12  # algorithmic TFLOP/s per GPU of the whole SVD job, from the timed region's wall clock
        roofline["svd_level"] = {"bound": "mfma", "unit_of_work": "one economy SVD, F = 14 m n^2 + 8 n^3", "achieved": achieved, "peak": 157.3,
                                 "unit": "TFLOP/s", "frac": achieved / 157.3,
                                 "note": "fp32-MFMA peak as the yardstick (SURVEY 8d); the streaming products run split-fp16 on the 16-bit matrix pipe (3 products per fp32 product) and as int8 digit products (8 / 9 per product), so this fraction does not bound what executes"}
        if two_level:
            roofline["streaming_kernels"] = {k: {"GBps": alg[k] / (classes[k]["avg_us"] * 1e-6) / 1e9, "frac_of_8TBps": alg[k] / (classes[k]["avg_us"] * 1e-6) / 8e12,
                                                 "avg_launch_us": classes[k]["avg_us"], "algorithmic_bytes_per_launch": alg[k]}
                                             for k in ("supgram", "supdate", "sgram") if classes[k]["launches"]}
        # HBM bytes one SVD moves: algorithmic (the streaming kernels' counters of this step) and measured (PMC, every kernel of the step)
        if two_level:
            alg_step = sum(alg[k] * classes[k]["launches"] for k in ("supgram", "supdate", "sgram"))
            roofline["hbm_bytes_per_svd"] = {"pmc_all_kernels": pmc_per_svd, "algorithmic_two_level_streaming_kernels": alg_step / B,
                                             "minimal_io": 4.0 * m * n + 4.0 * min(m, n) * (m + n + 1),
                                             "note": "pmc_all_kernels: (2 x FETCH_SIZE + WRITE_SIZE) summed over EVERY kernel of one step / batch, from the stored counter passes of "
                                                     "this library build (null when profiles/pmc_traffic.json belongs to another build); the doubling is calibrated for wide "
                                                     "streaming reads only (MI355X_MICROARCH.md, HBM)"}
        # ---- the configuration the timed steps ran in: two half-batch calls on two CU-masked streams ----
        if split_prof is not None:
            n8 = [h["supgram"]["launches"] for h in split_prof["halves"]]
            ms8 = [h["supgram"]["ms"] for h in split_prof["halves"]]
            Bh = [(B + 1) // 2, B // 2]
            wr_total = 1.0 * rows_j * 128 * 4 * split_prof["pairs"]["super_updates"]          # bytes written by all update launches of the step
            upd_l = sum(h["supgram"]["launches"] + h["supdate"]["launches"] for h in split_prof["halves"])
            bytes8 = [(1.0 * rows_j * 128 * 4 * (Bh[h] * (ns_ // 2)) + wr_total / max(1, upd_l)) * n8[h] for h in range(2)]   # algorithmic bytes of a half's class-8 launches
            ov = split_prof["supgram_ms"]
            roofline["split"] = {
                "what": "one extra untimed step profiled WITHOUT undoing the split (asvd_svd_set_profiling mode 2): HIP events of each half on its own stream; this is "
                        "the configuration `value` / `ms_per_step` were timed in",
                "wall_ms_profiled_split_step": split_wall_ms,
                "halves": [{"problems": Bh[h], "cus": 128, "classes_ms": {k: v["ms"] for k, v in split_prof["halves"][h].items()},
                            "launches": {k: v["launches"] for k, v in split_prof["halves"][h].items()},
                            "classes_sum_ms": sum(v["ms"] for v in split_prof["halves"][h].values()),
                            "supgram_avg_launch_us": (1e3 * ms8[h] / n8[h]) if n8[h] else None,
                            "supgram_GBps_of_this_half": (bytes8[h] / (ms8[h] * 1e-3) / 1e9) if ms8[h] else None} for h in range(2)],
                "supgram_both_halves": {"sum_ms_half0": ov["half0"], "sum_ms_half1": ov["half1"], "union_ms": ov["union"], "both_in_flight_ms": ov["both"],
                                        "algorithmic_bytes_per_step": sum(bytes8),
                                        "combined_GBps_over_union": (sum(bytes8) / (ov["union"] * 1e-3) / 1e9) if ov["union"] else None,
                                        "frac_of_8TBps": (sum(bytes8) / (ov["union"] * 1e-3) / 8e12) if ov["union"] else None,
                                        "note": "all fused update + Gram launches of both halves placed on one time axis (events against a common base event): "
                                                "bytes of both halves / the time AT LEAST ONE half was inside such a launch"},
                "kernel_trace": "profiles/r6_bench_kernel_stats_batch32_split_default.txt"}
        roofline["dominant_by_total_time"] = dom_all
        roofline["pairs"] = pairs_cnt
        roofline["sweep_wall_ms"] = sweep_ms
        roofline["sweep_rotated_pairs"] = sweep_rot
        roofline["classes"] = classes
        roofline["sweeps"] = [i.sweeps for i in infos]
        out = {
            "metric": "weight-matrix SVDs/sec (4096x4096 fp32)", "value": value, "unit": "SVD/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": prewarm_steps, "ms_per_step": 1e3 * dt / args.steps,
            "step_wall_ms": [1e3 * (b - a) for a, b in zip([t0] + step_marks[:-1], step_marks)], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 in, fp32 U / S / V out (the rank-r factors are emitted in fp16); inside: the Gram matrix of the Cholesky-QR EXACTLY from int8 digit products "
                          "(24-bit fixed point per column, nine products, int32 accumulation, fp64 combine: csrc/gram_i8.h), the Cholesky factorisation in fp64 MFMA, fp32 VALU for the "
                          "64x64 eigen-solves, the dense-sweep update + Gram as THREE fp16 MFMA products per fp32 product with power-of-two column scales (2^-22 relative per "
                          "product), the coupling snapshot and the long-side GEMM as EIGHT int8 digit products with exact int32 accumulation (operands rounded to 2^-25 of "
                          "their column / row maximum: csrc/snapshot_i8.h, csrc/nn_gemm_i8.h): fp32-equivalent by measurement, not by construction — per-column error against "
                          "fp64 and int64 arithmetic in tests/test_gpu_twolevel.py / test_gpu_gram_i8.py, whole-SVD parity on flat AND graded / clustered inputs in "
                          "tests/test_gpu_svd.py / test_gpu_families.py",
            "config": {"workload": f"{B} synthetic {m}x{n} fp32 Linears per GPU per step, abs_mean scaling (alpha 0.5), full SVD + rank-{r} truncation, factors emitted in fp16 (SURVEY 8d; the reference would emit the Linear's own dtype, svd_linear.py:102 - the cast is <0.1% of a step)",
                       "batch_per_gpu": B, "m": m, "n": n, "rank": r, "parallelism": f"independent matrices x{world}",
                       "batch_split_over_chip_halves": bool(any(i.split for i in timed_infos)),
                       "split_refused_device_shared_or_masked": bool(any(i.split_refused for i in timed_infos)),
                       "split_note": "asvd_svd_batched runs a batch of >= 4 problems with >= 3072 columns as two halves on CU-masked streams (128 CUs each, one host "
                                     "thread each; DESIGN.md 3.11) unless another process computes on the device: the flag above is what the TIMED steps did (path bits of the "
                                     "library).  `roofline` = a profiled step that runs UNSPLIT, every kernel alone on the whole chip (what profiles/*kernel_stats* with "
                                     "ASVD_SPLIT=0 shows); `roofline.split` = a profiled step in the timed configuration"},
            "roofline": roofline,
            "lib_sha256": lib_sha,
            "lib_build": "prebuilt in-tree asvd4llm_amd/libasvd_hip.so loaded with ctypes; bench.py never compiles (__graft_entry__.build() compiles only when a source is newer than the .so)",
        }
        if per_rank is not None:
            out["per_rank_svds_per_s"] = per_rank

    # ---- device inventory of the run (host-side gather on the gloo group): what the first multi-GPU record must carry to be read later ----
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": gpu_index, "name": props.name, "cus": props.multi_processor_count,
          "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "hostname_pid": f"{os.uname().nodename}:{os.getpid()}"}
    everyone = [me]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
    if rank == 0:
        try:
            rccl_v = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            rccl_v = None
        out["devices"] = {"visible_device_count": torch.cuda.device_count(), "rccl_version": rccl_v, "per_rank": everyone,
                          "host_group_backend": "gloo" if world > 1 else "none (single process)"}
        for h in out["roofline"].get("split", {}).get("halves", []):
            h["cus"] = props.multi_processor_count // 2

    # The line is complete from here on.  Everything below is an untimed extra; a watchdog prints the line and ends the process if an extra does not
    # come back (a collective that has never run on this node must not cost the weak-scaling number).  ONE printer: whoever takes the lock first.
    import threading
    line_lock = threading.Lock()
    state = {"printed": False}

    def print_line_once():
        with line_lock:
            if state["printed"]:
                return False
            state["printed"] = True
            if rank == 0:
                print(json.dumps(out), flush=True)
            return True

    def guarded(seconds, label, record_key):
        """threading.Timer that, when `label` overruns, records that under out[record_key], prints the line and exits the process"""
        def give_up():
            if rank == 0 and out is not None:
                with line_lock:
                    if not state["printed"]:
                        out[record_key] = {"model": sm, "error": f"{label} did not finish within {seconds:.0f} s: abandoned (the line above it is complete)", "abandoned": True}
            if print_line_once() or rank != 0:
                os._exit(0)
        t = threading.Timer(seconds, give_up)
        t.daemon = True
        t.start()
        return t

    sm = args.sharded_model if args.sharded_model != "auto" else "llama-2-7b"
    if world > 1:
        U = S = V = scales = outs = None   # drop the held outputs of the last timed step (3 GB)

    # ---- N > 1: configs[3] in the same line (untimed): the model's Linears LPT-sharded over the ranks + the sensitivity all-gather and the factor gather
    # on the DEVICE group — RCCL, created here, inside the watchdog; if it cannot be created the leg still runs, on gloo, and says so ----
    if world > 1 and sm != "none":
        watchdog = guarded(args.sharded_timeout_s, "the sharded-model leg (incl. creating the RCCL group)", "sharded_model")
        coll_note = None
        try:
            from asvd4llm_amd import parallel
            if args.dist_backend == "nccl":
                try:
                    if os.environ.get("ASVD_BENCH_FAIL_NCCL"):   # tests: the RCCL group cannot be had
                        raise RuntimeError("ASVD_BENCH_FAIL_NCCL is set")
                    import datetime
                    grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(60.0, args.sharded_timeout_s)))
                    probe = torch.ones(1, device=dev)
                    dist.all_reduce(probe, group=grp)   # communicators are created lazily: make the first RCCL call here, under the watchdog
                    torch.cuda.synchronize()
                    assert int(probe.item()) == world
                    parallel.set_group(grp)
                except Exception as e:   # noqa: BLE001 — fall back to the host group and say so
                    coll_note = f"RCCL group unavailable ({type(e).__name__}: {e})"[:200] + "; the leg ran on the gloo host group"
                    parallel.set_group(None)
            sharded = sharded_model_leg(sm, rank, world, dev)
            if coll_note and isinstance(sharded, dict):
                sharded["collective_backend_note"] = coll_note
        except Exception as e:   # noqa: BLE001 — the untimed extra must not cost the bench line
            sharded = {"model": sm, "error": f"{type(e).__name__}: {e}"[:300]}
            print(f"[bench] rank {rank}: sharded-model leg failed: {sharded['error']}", file=sys.stderr)
        watchdog.cancel()
        if rank == 0:
            with line_lock:
                if not state["printed"]:
                    out["sharded_model"] = sharded

    if rank == 0 and world == 1:
        # ---- batch-1 latency (BASELINE configs[1] says "single ... Linear"): one matrix alone, same path, median of 3 ----
        if not args.no_latency:
            lat = []
            for _ in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sc1 = ops.make_scale(stats[0], alpha=0.5)
                U1, S1, V1, i1 = ops.svd(mats[0], sc1)
                ops.truncate_split(U1, S1, V1, sc1, r, "UV", torch.float16)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t1)
            lat = sorted(lat[1:])
            out["latency_batch1_ms"] = 1e3 * lat[len(lat) // 2]
            out["svds_per_s_batch1"] = 1.0 / lat[len(lat) // 2]
            # BASELINE configs[1] reads "single ... Linear": the literal one-matrix figure travels inside `config` with the workload it belongs to
            out["config"]["latency_batch1_ms"] = out["latency_batch1_ms"]
            out["config"]["svds_per_s_batch1"] = out["svds_per_s_batch1"]
        # ---- K9 on the device (north_star "reconstructed W <= 1e-3 Frobenius"): |W - A B|_F / |W|_F of the emitted fp16 factors of one matrix of the
        # batch, by the tiled fp16-MFMA kernel fused with the difference reduction (asvd_reconstruct_err), timed with HIP events on its stream ----
        A_g, B_g, _ = outs[0]
        ops.reconstruct_err(mats[0], A_g, B_g)  # warm (workspace)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            k9_out = ops.reconstruct_err(mats[0], A_g, B_g)
        e1.record()
        torch.cuda.synchronize()
        k9_us = e0.elapsed_time(e1) * 1e3 / 5
        e2w2 = k9_out.cpu()
        k9 = {"recon_err_over_W_device": float((e2w2[0] / e2w2[1]).sqrt()), "us_per_call": k9_us, "flop": 2.0 * m * n * r,
              "TFLOPs": 2.0 * m * n * r / (k9_us * 1e-6) / 1e12, "frac_of_fp16_mfma_peak": 2.0 * m * n * r / (k9_us * 1e-6) / 2.5e15,
              "note": "pad + transpose of the factors, the GEMM + reduction kernel and the final sum: three launches per call, all inside the timed region"}
        out["k9_reconstruct"] = k9
        # ---- parity + CPU baseline (rank 0, N=1 only): the oracle pipeline on the box's host cores, bounded sample ----
        best_t, default_threads = None, torch.get_num_threads()
        if not args.no_cpu_baseline:
            from oracle import asvd_oracle as O

            def cpu_once(b=0):
                Wb, sb = mats[b].cpu(), O.make_scale(stats[b].cpu(), 0.5)
                t1 = time.perf_counter()
                ws = O.scaled_weight(Wb, sb)
                Uo, So, Vo = O.exact_svd(ws)
                Ao, Bo, _ = O.truncate_split(Uo, So, Vo, sb, r, "UV", torch.float16)
                return time.perf_counter() - t1, (So, Ao, Bo, Uo, Vo, ws, Wb, sb)

            # 1 warm-up, then a thread sweep (one run each), then >= 3 repetitions at the best thread count, median; bounded
            # to about --cpu_budget_s seconds of CPU work
            t_budget0 = time.perf_counter()
            _, ref = cpu_once()  # warm-up (MKL first-call cost), also the parity reference of problem 0
            sweep = {}
            ncpu = os.cpu_count() or default_threads
            cand = [t for t in (32, 64, 16, 128, 8) if t <= ncpu]
            if not cand:
                cand = [max(1, min(ncpu, default_threads))]  # small hosts: the sweep always holds at least the default thread count
            for t in cand:
                if time.perf_counter() - t_budget0 > 0.5 * args.cpu_budget_s and sweep:
                    break
                torch.set_num_threads(t)
                sweep[t] = cpu_once()[0]
            best_t = min(sweep, key=sweep.get)
            torch.set_num_threads(best_t)
            reps = [sweep[best_t]]
            while len(reps) < 3 or (len(reps) < args.cpu_reps and time.perf_counter() - t_budget0 < args.cpu_budget_s):
                reps.append(cpu_once()[0])
            reps.sort()
            tcpu = reps[len(reps) // 2]
            # what the reference literally calls (modules/svd_linear.py:65): randomized torch.svd_lowrank(q=rank), one run — timed, and its factors KEPT:
            # SURVEY 8c's secondary criterion compares the HIP path's rank-r error with the stock reference's on the same scaled matrix
            So, Ao, Bo, Uo, Vo, ws0, W0, s0 = ref
            t1 = time.perf_counter()
            torch.manual_seed(233)
            Ul, Sl, Vl = torch.svd_lowrank(ws0, q=r)
            t_lowrank = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": 1.0 / tcpu, "unit": "SVD/s", "cores": best_t, "kind": "port",
                                   "sample": f"oracle scale + torch.linalg.svd (gesdd) + truncate/split on ONE {m}x{n} matrix of the batch: 1 warm-up, thread sweep "
                                             f"{sorted(sweep)} (one run each), median of {len(reps)} runs at the best thread count",
                                   "seconds_per_svd": tcpu, "seconds_by_threads": {str(k): v for k, v in sorted(sweep.items())},
                                   "host_cpu_count": ncpu, "seconds_torch_svd_lowrank_q_rank": t_lowrank}
            r9 = int(m * n * 0.9) // (m + n)
            wsd = ws0.to(dev).double()
            ws_norm = float(wsd.norm())
            err_lowrank = float((wsd - (Ul.to(dev).double() * Sl.to(dev).double()) @ Vl.to(dev).double().T).norm())
            err_hip = float((wsd - (U[0][:, :r].double() * S[0][:r].double()) @ V[0][:, :r].double().T).norm())       # the HIP path's fp32 triplets, rank r
            err_oracle = float((wsd - (Uo[:, :r].to(dev).double() * So[:r].to(dev).double()) @ Vo[:, :r].to(dev).double().T).norm())
            del wsd

            def parity_of(b, refb):
                So_, Ao_, Bo_, _, _, _, Wb_, sb_ = refb
                A_b, B_b, _ = outs[b]
                rerr, rerr_scaled = O.recon_parity(A_b, B_b, Ao_, Bo_, Wb_, sb_)
                # the same spectrum in fp64 (CPU LAPACK on the oracle's own fp32 input): separates the path's distance from the exact answer from the fp32
                # oracle's (gesdd in fp32 is itself 1e-5 ... 7e-5 off on the smallest retained values of this family)
                S64_ = torch.linalg.svdvals(refb[5].double())
                e64 = float(((S[b].cpu().double() - S64_).abs() / S64_)[:r9].max())
                o64 = float(((So_.double() - S64_).abs() / S64_)[:r9].max())
                return {"problem": b, "half": 0 if b < (B + 1) // 2 else 1, "sigma_rel_err_top_r": O.sigma_rel_err(S[b].cpu(), So_, r9),
                        "sigma_rel_err_top_r_vs_fp64_svdvals": e64, "oracle_fp32_own_rel_err_vs_fp64": o64,
                        "recon_fro_err_rank%d_vs_oracle" % r: rerr, "recon_fro_err_scaled_norm": rerr_scaled, "sweeps": timed_infos[b].sweeps}

            p0 = parity_of(0, ref)
            # the LAST problem of the batch runs in the second half of a split call (the worker thread, the second CU-masked stream): compared with
            # the oracle directly, not only with the first half
            pl = parity_of(B - 1, cpu_once(B - 1)[1]) if B > 1 else None
            torch.set_num_threads(default_threads)
            # the same quantity for the ORACLE's factors, on the host in fp64: what an exact rank-r truncation leaves (the device figure above must not exceed it
            # by more than the contract's 1e-3)
            oracle_err = float(((W0.double() - Ao.double() @ Bo.double()).norm() / W0.double().norm()).item())
            problems = [p0] + ([pl] if pl else [])
            out["parity"] = {"sigma_rel_err_top_r": max(q["sigma_rel_err_top_r"] for q in problems), "r": r9,
                             "recon_fro_err_rank512_vs_oracle": max(q["recon_fro_err_rank%d_vs_oracle" % r] for q in problems),
                             "recon_fro_err_scaled_norm": max(q["recon_fro_err_scaled_norm"] for q in problems),
                             "problems": problems, "problems_note": "worst of the first problem of the batch (first half of a split call) and the last (second half); outputs of the LAST TIMED step",
                             "timed_call_path": {"split": timed_infos[0].split, "reduced": timed_infos[0].reduced, "reduce_fallback": timed_infos[0].reduce_fallback, "gram_retry": any(i.gram_retry for i in timed_infos),
                                                 "plain_retry": timed_infos[0].plain_retry},
                             "truncation_err_over_W_device_k9": k9["recon_err_over_W_device"], "truncation_err_over_W_oracle_fp64": oracle_err,
                             "vs_stock_svd_lowrank": {"criterion": "SURVEY 8c secondary: |Ws - (rank-r of the HIP path)|_F <= |Ws - torch.svd_lowrank(Ws, q=r)|_F (1 + 1e-5); the exact "
                                                                   "truncation can only be better than the randomized one the reference calls (svd_linear.py:65)",
                                                      "rank": r, "err_hip_over_Ws": err_hip / ws_norm, "err_svd_lowrank_over_Ws": err_lowrank / ws_norm,
                                                      "err_oracle_over_Ws": err_oracle / ws_norm, "ok": bool(err_hip <= err_lowrank * (1 + 1e-5))},
                             "vs_stock_svd_lowrank_ok": bool(err_hip <= err_lowrank * (1 + 1e-5)),
                             "ok": bool(all(q["sigma_rel_err_top_r"] <= 1e-4 and q["recon_fro_err_rank%d_vs_oracle" % r] <= 1e-3 for q in problems)),
                             "tolerance": {"sigma": 1e-4, "recon": 1e-3}}
            del ref
        # ---- configs[2] in the same line (untimed, bounded by the same watchdog as the N > 1 leg): the whole model's decomposition on this GPU
        # ("full_model": the second component of BASELINE.json's metric), then per-shape parity against the CPU oracle and the CPU reference it
        # divides — the oracle pipeline timed ONCE per distinct shape at the best thread count found above, times the number of Linears of that shape ----
        if sm != "none":
            watchdog = guarded(args.sharded_timeout_s, "the full-model leg", "full_model")
            fm_samples = []
            try:
                U = S = V = scales = outs = None   # the profiled step's outputs (3 GB)
                sharded = sharded_model_leg(sm, rank, world, dev, samples=fm_samples)
                if best_t is not None and fm_samples and "error" not in sharded:
                    sharded.update(full_model_parity(fm_samples, sharded["model"], dev, best_t, sharded.get("decompose_s")))
                    torch.set_num_threads(default_threads)
            except Exception as e:   # noqa: BLE001 — the untimed extra must not cost the bench line
                sharded = {"model": sm, "error": f"{type(e).__name__}: {e}"[:300]}
                print(f"[bench] full-model leg failed: {sharded['error']}", file=sys.stderr)
            watchdog.cancel()
            with line_lock:
                if not state["printed"]:
                    out["full_model"] = sharded
    print_line_once()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
