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


# The process group every collective of this module runs on.  None = the default group (what asvd.py initialises: "nccl" = RCCL on GPUs,
# "gloo" in CPU tests).  bench.py keeps the DEFAULT group on gloo — rendezvous, timing barriers and the MAX-reduce of the weak-scaling number
# never touch RCCL — and points GROUP at a separate RCCL group for the sharded-model leg only (set_group), inside that leg's watchdog.
GROUP = None


def set_group(group):
    global GROUP
    GROUP = group


def backend():
    return dist.get_backend(GROUP)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(GROUP), dist.get_world_size(GROUP)
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


# ---- cost model of the ppl sweep (what the LPT shard of calib_sensitivity_ppl balances) -----------------------------------------------
# One layer of the sweep = one factorisation + ONE batched suffix pass per calibration sample (sweep_eval.PrefixCachedEvaluator): the blocks
# in FRONT of the swapped layer are replayed from cached hidden states, so a layer in block i of an N-block model costs the N - i blocks behind
# it plus the head, times the number of candidate ratios (they run as one batch), and lm_head costs its own GEMM only.  On Llama-2-7B with
# n_calib 32 the forwards are 755 of the 761 s (profiles/r4_e2e_llama2_7b_ncalib32.json): balancing the shard on SVD flops alone says nothing
# about the stage that is 99 % of the wall-clock.  Rates are the measured ones of round 4 (one MI355X): 826 TFLOP/s sustained through the
# fp16 forwards of the sweep, 109 TFLOP/s (full-SVD flop count) through the factorisations.
SWEEP_FORWARD_FLOPS_PER_S = 826e12
SWEEP_SVD_FLOPS_PER_S = 109e12


def sweep_layer_costs(layers, n_blocks, block_params, head_params, n_ratios, n_samples, seqlen, attn_flops_per_token_block=0.0):
    """Predicted seconds of every layer's share of the ppl sweep.  layers: [(out_features, in_features, where)] in traversal order with
    where = block index (0-based) for a Linear inside decoder block `where`, "after" for one that runs behind the last block (lm_head),
    "before" for one in front of the blocks (nothing can be replayed: full forward).  block_params / head_params: Linear parameters of one
    decoder block / of the layers behind the blocks (2 flop per parameter and token)."""
    tokens = float(n_samples) * float(seqlen) * float(n_ratios)
    per_block = 2.0 * block_params + attn_flops_per_token_block
    head = 2.0 * head_params
    costs = []
    for out_f, in_f, where in layers:
        if where == "after":
            fwd = head
        elif where == "before":
            fwd = n_blocks * per_block + head
        else:
            fwd = (n_blocks - int(where)) * per_block + head
        costs.append(svd_flops(out_f, in_f) / SWEEP_SVD_FLOPS_PER_S + tokens * fwd / SWEEP_FORWARD_FLOPS_PER_S)
    return costs


def sweep_costs_for_model(model, linears, n_ratios, n_samples, seqlen, prefix_cached=True):
    """sweep_layer_costs for the nn.Linears of `model` (linears: [(module, info)] as sensitivity.collect_linear_info yields them): the block
    structure is read the way the evaluator reads it (sweep_eval.find_decoder_blocks); a model without a block list falls back to SVD flops."""
    import torch.nn as nn
    from .sweep_eval import find_decoder_blocks
    blocks_name, blocks = find_decoder_blocks(model)
    if blocks is None:
        return [svd_flops(l.out_features, l.in_features) for l, _ in linears]
    prefix = blocks_name + "."
    n_blocks = len(blocks)
    block_params = sum(m.weight.numel() for m in blocks[0].modules() if isinstance(m, nn.Linear))
    where, head_params = [], 0
    for l, info in linears:
        name = info["full_name"]
        if name.startswith(prefix):
            where.append(int(name[len(prefix):].split(".")[0]))
        else:
            # the capture pass of the evaluator decides "before" / "after" from the call order; by name: heads come after, projections in
            # front of the blocks (OPT project_in) before
            w = "before" if ("project_in" in name or "embed" in name) else "after"
            where.append(w)
            if w == "after":
                head_params += l.weight.numel()
    hidden = getattr(getattr(model, "config", None), "hidden_size", None) or 0
    attn = 4.0 * hidden * (seqlen / 2.0)   # QK^T and PV under a causal mask, per token and block
    if not prefix_cached:   # --no_fused_sweep: every evaluation is a full forward
        where = ["before"] * len(where)
    return sweep_layer_costs([(l.out_features, l.in_features, w) for (l, _), w in zip(linears, where)], n_blocks, block_params, head_params,
                             n_ratios, n_samples, seqlen, attn)


def load_balance(costs, owner, world_size):
    """max over ranks / mean over ranks of the summed cost"""
    load = [0.0] * world_size
    for c, o in zip(costs, owner):
        load[o] += c
    mean = sum(load) / world_size
    return (max(load) / mean) if mean > 0 else 1.0


def _comm_device():
    return torch.device("cuda", torch.cuda.current_device()) if backend() == "nccl" else torch.device("cpu")


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=GROUP)


def cache_exists(path):
    """ONE answer for all ranks: rank 0 looks, every rank gets what it saw.  (With a shared working directory a rank that looks for itself
    can see a file another rank created a moment ago and take the load branch while its peers take the compute branch — and the compute
    branch may hold a collective.)"""
    if not (dist.is_available() and dist.is_initialized()) or world()[1] == 1:
        return os.path.exists(path)
    flag = torch.tensor([1 if (world()[0] == 0 and os.path.exists(path)) else 0], dtype=torch.int32, device=_comm_device())
    dist.broadcast(flag, src=0, group=GROUP)
    return bool(int(flag.item()))


def save_cache(obj, path):
    """`.pt` cache files are written by rank 0 only, to a temporary name that is renamed into place (a reader sees the complete file or
    none), and every rank learns whether that worked before it goes on: a failure on rank 0 (disk full, permissions) is raised on EVERY
    rank instead of leaving the others in a barrier.  The caches assume ONE working directory shared by all ranks (single node)."""
    rank, ws = world()
    err = None
    if rank == 0:
        try:
            d = os.path.dirname(path)
            if d:
                os.makedirs(d, exist_ok=True)
            tmp = f"{path}.tmp.{os.getpid()}"
            torch.save(obj, tmp)
            os.replace(tmp, path)
        except Exception as e:  # noqa: BLE001 — re-raised below, on all ranks
            err = e
    if ws > 1:
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=_comm_device())
        dist.broadcast(ok, src=0, group=GROUP)
        if int(ok.item()) == 0 and err is None:
            err = RuntimeError(f"rank 0 could not write the cache file {path} (see its stderr)")
    if err is not None:
        raise err


def load_cache(path):
    """torch.load of a cache file every rank is about to read; a rank that does not see it (no shared working directory) gets a clear error"""
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: rank 0 reported this cache file but rank {world()[0]} cannot see it — the .pt caches need ONE working "
                                "directory shared by all ranks (or run with use_cache=False)")
    return torch.load(path, map_location="cpu")


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
    if backend() == "nccl":
        dist.all_gather_into_tensor(out, buf, group=GROUP)
    else:
        dist.all_gather(list(out.unbind(0)), buf, group=GROUP)
    out = out.cpu()
    full = {}
    for n in names:
        o, slot = slots[n]
        full[n] = {r: float(out[o, slot * len(ratios) + j]) for j, r in enumerate(ratios)}
        assert all(float(out[o, stride + slot * len(ratios) + j]) == 1.0 for j in range(len(ratios))), f"missing sensitivity for {n} (owner rank {o})"
    return full


def _wire_of(mod, raw):
    """(header, payload tensors) of one decomposed layer: int64 [kind, rank, has_bias, out, in], then ALinear.weight, BLinear.weight (kind 1 =
    SVDLinear) or weight (kind 0 = plain nn.Linear: the reference's random-Linear fallback after a failed factorisation), then the bias"""
    from .modules.svd_linear import SVDLinear
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
    return hdr, payload


def _module_from_wire(kind, r, has_bias, out_f, in_f, ts, dtype):
    import torch.nn as nn
    from .modules.svd_linear import SVDLinear
    bias_t = ts[-1] if has_bias else None
    if kind == 1:
        return SVDLinear._from_factors(ts[0], ts[1], bias_t, r)
    new = nn.Linear(in_f, out_f, bias=bool(has_bias)).to(dtype)
    new.weight.data = ts[0]
    if has_bias:
        new.bias.data = bias_t
    return new


def _wire_shapes(kind, r, has_bias, out_f, in_f):
    shapes = [(out_f, r), (r, in_f)] if kind == 1 else [(out_f, in_f)]
    if has_bias:
        shapes.append((out_f,))
    return shapes


def _p2p_all(ops):
    """ops: (fn, tensor, peer, tag) tuples in posting order.  Posted in ROUNDS that hold at most one message per peer — the c-th message of
    every peer goes into round c — and every round is one batch_isend_irecv (RCCL: one group call): the messages of different peers travel
    side by side (rank 0 drives all its xGMI links at once), the messages of one peer keep their order, and no group holds more than one
    operation per peer (older RCCL / NCCL releases refuse several sends to the same peer inside one group).  A round ends when all its
    transfers have; both ends of a pair post their c-th message in their c-th round, so the rounds need no global agreement."""
    queues = {}
    for op in ops:
        queues.setdefault(op[2], []).append(op)
    depth = max((len(q) for q in queues.values()), default=0)
    for c in range(depth):
        batch = [dist.P2POp(q[c][0], q[c][1], q[c][2], group=GROUP, tag=q[c][3]) for q in queues.values() if c < len(q)]
        for req in dist.batch_isend_irecv(batch):
            req.wait()


def exchange_factors(items, owner, mode="rank0"):
    """Complete the model after a sharded decomposition (binary_search_truncation_rank with layers LPT-sharded over ranks).

    items: ordered list of (full_name, father_module, child_name, raw_linear) for every layer the search decided to factorise —
    identical on all ranks; `owner[full_name]` decomposed it and holds the new module under father.child_name, every other rank
    still holds the raw nn.Linear there.
      mode "all"   : the owner broadcasts its factors, every rank ends with the complete compressed model;
      mode "rank0" : the owners send them to rank 0 only, point to point (each transfer crosses one xGMI link, nothing is relayed around a
                     ring), ALL OWNERS AT ONCE: in every round rank 0 posts one receive per owner and every owner one send (_p2p_all), so
                     the seven links into rank 0 carry their shards concurrently (SURVEY 5 sizes the ~12 GB of Llama-2-7B for all links
                     driven together; a layer-by-layer blocking send / recv used one link at a time).  Two phases: the headers (rank 0
                     cannot know whether an owner fell back to a plain Linear), then the payloads;
      mode "none"  : nothing moves.
    Wire format per layer: see _wire_of.  Returns the number of layers this rank received."""
    if mode == "none" or not (dist.is_available() and dist.is_initialized()):
        return 0
    rank, ws = world()
    dev = _comm_device()
    received = 0
    if mode == "rank0":
        moving = [(i, it) for i, it in enumerate(items) if owner[it[0]] != 0]
        if rank != 0:
            moving = [(i, it) for i, it in moving if owner[it[0]] == rank]
        if not moving:
            return 0
        # round 1: headers
        hdrs, wires, ops = {}, {}, []
        for i, (full_name, father, child, raw) in moving:
            if rank == 0:
                hdrs[i] = torch.empty(5, dtype=torch.int64, device=dev)
                ops.append((dist.irecv, hdrs[i], owner[full_name], 4 * i))
            else:
                h, payload = _wire_of(getattr(father, child), raw)
                hdrs[i] = torch.tensor(h, dtype=torch.int64, device=dev)
                wires[i] = [t.to(dev).contiguous() for t in payload]
                ops.append((dist.isend, hdrs[i], 0, 4 * i))
        _p2p_all(ops)
        # round 2: payloads
        ops, metas = [], {}
        for i, (full_name, father, child, raw) in moving:
            if rank == 0:
                kind, r, has_bias, out_f, in_f = (int(v) for v in hdrs[i].tolist())
                assert (out_f, in_f) == (raw.out_features, raw.in_features), f"factor exchange out of step at {full_name}"
                ts = [torch.empty(shp, dtype=raw.weight.dtype, device=dev) for shp in _wire_shapes(kind, r, has_bias, out_f, in_f)]
                metas[i] = (kind, r, has_bias, out_f, in_f, ts)
                for j, t in enumerate(ts):
                    ops.append((dist.irecv, t, owner[full_name], 4 * i + 1 + j))
            else:
                for j, t in enumerate(wires[i]):
                    ops.append((dist.isend, t, 0, 4 * i + 1 + j))
        _p2p_all(ops)
        if rank == 0:
            for i, (full_name, father, child, raw) in moving:
                kind, r, has_bias, out_f, in_f, ts = metas[i]
                wdev = raw.weight.device
                ts = [(t.to(wdev) if wdev.type != "cpu" or dev.type == "cpu" else t) for t in ts]
                setattr(father, child, _module_from_wire(kind, r, has_bias, out_f, in_f, ts, raw.weight.dtype))
                received += 1
        return received

    # mode "all": one broadcast per tensor, layer by layer (a broadcast is a collective of the whole group: they serialise anyway)
    for full_name, father, child, raw in items:
        src = owner[full_name]
        dtype, wdev = raw.weight.dtype, raw.weight.device
        if rank == src:
            hdr, payload = _wire_of(getattr(father, child), raw)
            dist.broadcast(torch.tensor(hdr, dtype=torch.int64, device=dev), src=src, group=GROUP)
            for t in payload:
                dist.broadcast(t.to(dev).contiguous(), src=src, group=GROUP)
            continue
        h = torch.empty(5, dtype=torch.int64, device=dev)
        dist.broadcast(h, src=src, group=GROUP)
        kind, r, has_bias, out_f, in_f = (int(v) for v in h.tolist())
        assert (out_f, in_f) == (raw.out_features, raw.in_features), f"factor exchange out of step at {full_name}"
        ts = []
        for shp in _wire_shapes(kind, r, has_bias, out_f, in_f):
            t = torch.empty(shp, dtype=dtype, device=dev)
            dist.broadcast(t, src=src, group=GROUP)
            ts.append(t.to(wdev) if wdev.type != "cpu" or dev.type == "cpu" else t)
        setattr(father, child, _module_from_wire(kind, r, has_bias, out_f, in_f, ts, dtype))
        received += 1
    return received
