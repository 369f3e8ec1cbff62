"""Calibration statistics — MI355X implementation of act_aware_utils.py:47-95 behind the same signature.

`calib_input_distribution(model, calib_loader, method, use_cache=True)` attaches `module.scaling_diag_matrix` ([C_in], in the
activation dtype, SUM over calibration batches of the per-batch |x| column mean / running max) to every nn.Linear and
writes the same cache file (`cache/{model_id with '/'->'_'}_calib_input_distribution_{method}.pt`, dict name -> tensor).
The hook body is one call into libasvd_hip.so (asvd_absstat_accum): a single pass over the [T, C] activation with 16-byte
coalesced loads and fp32 partial sums, instead of the reference's abs -> mean -> add chain with a [T, C] temporary."""
import os

import torch
import torch.nn as nn
from tqdm import tqdm

from . import ops


def _hook_factory(method):
    # Linears called with the SAME input tensor in one forward (q/k/v_proj, gate/up_proj) share the pass over X: the partial column
    # statistics of the last input are kept (with the tensor itself, so its address cannot be recycled) and only the tiny finalize
    # runs for the followers — X is read once per distinct input instead of once per Linear (the reference re-reads it every time)
    last = {"key": None, "x": None, "work": None}

    def hook(module, input, output):
        x = input[0].detach()
        if x.dim() >= 2:
            lead = 1
            for d in x.shape[:-2]:
                lead *= d
            if lead != 1:
                # the reference's `.view(-1)` would yield B*C elements and break the add: batch must be 1
                raise ValueError(f"calibration hook needs batch size 1, got input of shape {tuple(x.shape)}")
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        acc = module.scaling_diag_matrix
        if not torch.is_tensor(acc):  # python int 0 start (act_aware_utils.py:80)
            acc = torch.zeros(x2.shape[1], dtype=x2.dtype, device=x2.device)
            module.scaling_diag_matrix = acc
        key = (x2.data_ptr(), tuple(x2.shape), x2.stride(0), x2.dtype, x._version)
        if last["key"] != key:
            last["key"], last["x"], last["work"] = key, x2, ops.absstat_partial(x2, method)
        ops.absstat_finalize(last["work"], x2.shape[0], x2.shape[1], acc, method)

    def release():
        last["key"] = last["x"] = last["work"] = None

    hook.release = release
    return hook


def _allreduce_statistics(model, method):
    """--shard_calib: combine the per-rank accumulators (SURVEY.md 8e "optional"): sum for abs_mean, max for abs_max — ONE collective
    over a flat buffer of all [C] vectors.  The sum runs in fp32 and is rounded once to the activation dtype (the replicated pass adds
    batch by batch in that dtype: equal up to that rounding, which is why this is opt-in)."""
    import torch.distributed as dist
    from . import parallel
    # the buffer layout comes from the model's STRUCTURE, identical on every rank: a rank that ran no sample (fewer samples than ranks) or
    # whose forward never reached a Linear still holds the python int 0 there and contributes zeros (the neutral element of both the sum
    # and the max of absolute values)
    mods = [m for _, m in model.named_modules() if isinstance(m, nn.Linear)]
    if not mods:
        return
    dev = parallel._comm_device()
    for m in mods:
        if not torch.is_tensor(m.scaling_diag_matrix):
            m.scaling_diag_matrix = torch.zeros(m.in_features, dtype=m.weight.dtype, device=m.weight.device)
    flat = torch.cat([m.scaling_diag_matrix.float().reshape(-1) for m in mods]).to(dev)
    dist.all_reduce(flat, op=dist.ReduceOp.MAX if "abs_max" in method else dist.ReduceOp.SUM, group=parallel.GROUP)
    off = 0
    for m in mods:
        n = m.scaling_diag_matrix.numel()
        m.scaling_diag_matrix = flat[off:off + n].to(m.scaling_diag_matrix.device).to(m.scaling_diag_matrix.dtype)
        off += n


@torch.no_grad()
def calib_input_distribution(model, calib_loader, method, use_cache=True, shard_samples=False):
    """shard_samples (additive; asvd.py --shard_calib): with torch.distributed initialised, rank r runs the hook pass over the calibration
    samples i = r (mod world size) only and the accumulators are all-reduced — the default is the replicated pass (bit-identical
    statistics on every rank, no collective)."""
    model_id = model.config._name_or_path
    cache_file = f"cache/{model_id.replace('/','_')}_calib_input_distribution_{method}.pt"
    from . import parallel
    if use_cache and parallel.cache_exists(cache_file):
        all_scaling_diag_matrix = parallel.load_cache(cache_file)
        for name, module in model.named_modules():
            if isinstance(module, nn.Linear):
                module.scaling_diag_matrix = all_scaling_diag_matrix[name].to(module.weight.device)
        return
    model.eval()
    if "abs_mean" not in method and "abs_max" not in method:
        return  # the reference hook does nothing for other methods
    hook = _hook_factory(method)
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear):
            module.scaling_diag_matrix = 0
            module.register_forward_hook(hook)

    rank, ws = parallel.world()
    shard = bool(shard_samples) and ws > 1
    for i, batch in enumerate(tqdm(calib_loader, disable=(rank != 0))):
        if shard and i % ws != rank:
            continue
        batch = {k: v.to(model.device) for k, v in batch.items()}
        model(**batch)
        hook.release()
    if shard:
        _allreduce_statistics(model, method)

    all_scaling_diag_matrix = {}
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear):
            module._forward_hooks.clear()  # the reference clears ALL forward hooks of every Linear (act_aware_utils.py:93)
            all_scaling_diag_matrix[name] = module.scaling_diag_matrix
    parallel.save_cache(all_scaling_diag_matrix, cache_file)  # rank 0 writes (temporary name + rename), every rank waits for it


def calib_fisher_info(model, calib_loader, use_cache=True):
    """Fisher scaling statistics (act_aware_utils.py:8-44, `--scaling_method fisher*`): per Linear,
    fisher_info = sqrt( mean over batches of  weight.grad.pow(2).mean(0) ).  The backward pass is ordinary PyTorch model execution;
    the per-input-channel statistic of the gradient is the sq_mean mode of the hook kernel (asvd_absstat_accum)."""
    model_id = model.config._name_or_path
    cache_file = f"cache/{model_id.replace('/','_')}_calib_fisher_info.pt"
    from . import parallel
    if use_cache and parallel.cache_exists(cache_file):
        all_fisher_info = parallel.load_cache(cache_file)
        for name, module in model.named_modules():
            if isinstance(module, nn.Linear):
                module.fisher_info = all_fisher_info[name].to(module.weight.device)
        return
    model.eval()
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear):
            module.fisher_info = 0
    for batch in tqdm(calib_loader):
        input_ids = batch["input_ids"][:, :-1].to(model.device)
        labels = batch["input_ids"][:, 1:].to(model.device)
        out = model(input_ids=input_ids, labels=labels)
        out[0].backward()
        for name, module in model.named_modules():
            if isinstance(module, nn.Linear):
                g = module.weight.grad.detach()
                if not torch.is_tensor(module.fisher_info):
                    module.fisher_info = torch.zeros(g.shape[1], dtype=g.dtype, device=g.device)
                ops.absstat_accum(g if g.stride(1) == 1 else g.contiguous(), module.fisher_info, "sq_mean")
        model.zero_grad()
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear):
            module.fisher_info = module.fisher_info.div(len(calib_loader)).sqrt()
    all_fisher_info = {}
    for name, module in model.named_modules():
        if isinstance(module, nn.Linear):
            module._forward_hooks.clear()
            all_fisher_info[name] = module.fisher_info
    parallel.save_cache(all_fisher_info, cache_file)
