"""Multi-GPU layer sharding for the sweep / decomposition (new design; the reference has no distributed path).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests).  Every Linear's
scaling, SVD, truncation and (layer, ratio) -> ppl evaluation is independent, so layers are sharded by an LPT (longest
processing time first) assignment on the SVD flop estimate; the only exchange is ONE small all-gather of the per-layer
sensitivities (<= 29 x 6 fp32 per rank for Llama-2-7B) before the — replicated, deterministic — binary search."""
import math

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


def allgather_sensitivities(local, names, ratios, owner):
    """local: {name: {ratio: float}} for the layers this rank owns.  names/ratios/owner are identical on all ranks.
    Returns the complete dict {name: {ratio: float}} in `names` order on every rank.
    Wire format: one fixed-stride fp64 buffer (python floats survive bit-exactly) [max_local * len(ratios)] per rank, NaN padded."""
    rank, ws = world()
    if ws == 1:
        return {n: dict(local[n]) for n in names}
    slots = {}
    counts = [0] * ws
    for i, n in enumerate(names):
        slots[n] = (owner[i], counts[owner[i]])
        counts[owner[i]] += 1
    stride = max(counts) * len(ratios)
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.full((max(stride, 1),), float("nan"), dtype=torch.float64, device=dev)
    for n, (o, slot) in slots.items():
        if o == rank:
            for j, r in enumerate(ratios):
                buf[slot * len(ratios) + j] = float(local[n][r])
    out = torch.empty((ws, buf.numel()), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf) if backend == "nccl" else dist.all_gather(list(out.unbind(0)), buf)
    out = out.cpu()
    full = {}
    for n in names:
        o, slot = slots[n]
        full[n] = {r: float(out[o, slot * len(ratios) + j]) for j, r in enumerate(ratios)}
        assert not any(math.isnan(v) for v in full[n].values()), f"missing sensitivity for {n}"
    return full
