"""Multi-GPU layer sharding for the sweep / decomposition (new design; the reference has no distributed path).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests).  Every Linear's
scaling, SVD, truncation and (layer, ratio) -> ppl evaluation is independent, so layers are sharded by an LPT (longest
processing time first) assignment on the SVD flop estimate; the only exchange is ONE small all-gather of the per-layer
sensitivities (<= 29 x 6 fp32 per rank for Llama-2-7B) before the — replicated, deterministic — binary search."""
import math
import os

import torch
import torch.distributed as dist


def svd_flops(out_features, in_features):
    """Golub-Van Loan economy SVD flop count used for balancing and roofline accounting: 14 m n^2 + 8 n^3, m >= n"""
    m, n = max(out_features, in_features), min(out_features, in_features)
    return 14.0 * m * n * n + 8.0 * n ** 3


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def lpt_assign(costs, world_size):
    """costs: list of floats (one per unit, in traversal order).  Returns owner[i] in [0, world_size): greedy longest-first
    onto the least-loaded rank, ties by lower rank — deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def _comm_device():
    backend = dist.get_backend()
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def cache_exists(path):
    """ONE answer for all ranks: rank 0 looks, every rank gets what it saw.  (With a shared working directory a rank that looks for itself
    can see a file another rank created a moment ago and take the load branch while its peers take the compute branch — and the compute
    branch may hold a collective.)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return os.path.exists(path)
    flag = torch.tensor([1 if (dist.get_rank() == 0 and os.path.exists(path)) else 0], dtype=torch.int32, device=_comm_device())
    dist.broadcast(flag, src=0)
    return bool(int(flag.item()))


def save_cache(obj, path):
    """`.pt` cache files are written by rank 0 only, to a temporary name that is renamed into place (a reader sees the complete file or
    none), and every rank waits until the file is there before it goes on."""
    rank, _ = world()
    if rank == 0:
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save(obj, tmp)
        os.replace(tmp, path)
    barrier()


def allgather_sensitivities(local, names, ratios, owner):
    """local: {name: {ratio: float}} for the layers this rank owns.  names/ratios/owner are identical on all ranks.
    Returns the complete dict {name: {ratio: float}} in `names` order on every rank.
    Wire format: one fixed-stride fp64 buffer per rank, [max_local * len(ratios)] values followed by as many presence flags
    (python floats survive bit-exactly; NaN / inf perplexities are legitimate values and pass through unchanged).
    With a process group of size 1 the collective still runs (RCCL world-size-1 smoke path); without one it is a copy."""
    if not (dist.is_available() and dist.is_initialized()):
        return {n: dict(local[n]) for n in names}
    rank, ws = world()
    slots = {}
    counts = [0] * ws
    for i, n in enumerate(names):
        slots[n] = (owner[i], counts[owner[i]])
        counts[owner[i]] += 1
    stride = max(max(counts) * len(ratios), 1)
    dev = _comm_device()
    host = torch.zeros((2 * stride,), dtype=torch.float64)
    for n, (o, slot) in slots.items():
        if o == rank:
            for j, r in enumerate(ratios):
                host[slot * len(ratios) + j] = float(local[n][r])
                host[stride + slot * len(ratios) + j] = 1.0
    buf = host.to(dev)
    out = torch.empty((ws, buf.numel()), dtype=torch.float64, device=dev)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, buf)
    else:
        dist.all_gather(list(out.unbind(0)), buf)
    out = out.cpu()
    full = {}
    for n in names:
        o, slot = slots[n]
        full[n] = {r: float(out[o, slot * len(ratios) + j]) for j, r in enumerate(ratios)}
        assert all(float(out[o, stride + slot * len(ratios) + j]) == 1.0 for j in range(len(ratios))), f"missing sensitivity for {n} (owner rank {o})"
    return full


def exchange_factors(items, owner, mode="rank0"):
    """Complete the model after a sharded decomposition (binary_search_truncation_rank with layers LPT-sharded over ranks).

    items: ordered list of (full_name, father_module, child_name, raw_linear) for every layer the search decided to factorise —
    identical on all ranks; `owner[full_name]` decomposed it and holds the new module under father.child_name, every other rank
    still holds the raw nn.Linear there.
      mode "all"   : the owner broadcasts its factors, every rank ends with the complete compressed model;
      mode "rank0" : the owner sends them to rank 0 only (point-to-point: each transfer crosses one xGMI link, nothing is relayed
                     around a ring) — rank 0 exports / evaluates, the other ranks keep their shard;
      mode "none"  : nothing moves.
    Wire format per layer: int64 header [kind, rank, has_bias, out, in] (kind 1 = SVDLinear, 0 = plain nn.Linear: the reference's
    random-Linear fallback after a failed factorisation), then ALinear.weight, BLinear.weight (or weight), then bias.
    Returns the number of layers this rank received."""
    if mode == "none" or not (dist.is_available() and dist.is_initialized()):
        return 0
    import torch.nn as nn
    from .modules.svd_linear import SVDLinear
    rank, ws = world()
    dev = _comm_device()
    received = 0

    def xfer(t, src, sending):
        if mode == "all":
            dist.broadcast(t, src=src)
        elif sending:
            dist.send(t, dst=0)
        else:
            dist.recv(t, src=src)

    for full_name, father, child, raw in items:
        src = owner[full_name]
        if mode == "rank0" and (src == 0 or rank not in (0, src)):
            continue
        sending = rank == src
        dtype, wdev = raw.weight.dtype, raw.weight.device
        if sending:
            mod = getattr(father, child)
            if isinstance(mod, SVDLinear):
                bias = mod.ALinear.bias
                hdr = [1, mod.truncation_rank, int(bias is not None), raw.out_features, raw.in_features]
                payload = [mod.ALinear.weight.data, mod.BLinear.weight.data]
            else:
                bias = mod.bias
                hdr = [0, 0, int(bias is not None), raw.out_features, raw.in_features]
                payload = [mod.weight.data]
            if bias is not None:
                payload.append(bias.data)
            h = torch.tensor(hdr, dtype=torch.int64, device=dev)
            xfer(h, src, True)
            for t in payload:
                xfer(t.to(dev).contiguous(), src, True)
            continue
        h = torch.empty(5, dtype=torch.int64, device=dev)
        xfer(h, src, False)
        kind, r, has_bias, out_f, in_f = (int(v) for v in h.tolist())
        assert (out_f, in_f) == (raw.out_features, raw.in_features), f"factor exchange out of step at {full_name}"
        shapes = [(out_f, r), (r, in_f)] if kind == 1 else [(out_f, in_f)]
        if has_bias:
            shapes.append((out_f,))
        ts = []
        for shp in shapes:
            t = torch.empty(shp, dtype=dtype, device=dev)
            xfer(t, src, False)
            ts.append(t.to(wdev) if wdev.type != "cpu" or dev.type == "cpu" else t)
        bias_t = ts[-1] if has_bias else None
        if kind == 1:
            new = SVDLinear._from_factors(ts[0], ts[1], bias_t, r)
        else:
            new = nn.Linear(in_f, out_f, bias=bool(has_bias)).to(dtype)
            new.weight.data = ts[0]
            if has_bias:
                new.bias.data = bias_t
        setattr(father, child, new)
        received += 1
    return received
