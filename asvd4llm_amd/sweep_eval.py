"""Prefix-cached sweep evaluator for calib_sensitivity_ppl.

The reference sweep (sensitivity.py:43-59) swaps ONE Linear, runs `evaluate_perplexity` (evaluate_utils.py:90-115: a full model
forward per calibration sample), restores the Linear, and repeats for every (layer, ratio): 1350 x n_calib full forwards on
Llama-2-7B.  Every one of those forwards recomputes the part of the network in front of the swapped layer, which is the
UNMODIFIED model every time (the sweep restores the raw Linear before moving on, sensitivity.py:59).

This evaluator runs the unmodified model once per calibration sample, keeps the hidden states entering each decoder block
resident in HBM (n_calib x n_blocks x T x C fp16: 17 GB for Llama-2-7B at n_calib=32 — sized for 288 GB, with a stride
fallback when memory is short), and evaluates a swapped layer that lives in block i by running only blocks i..N-1 and the
head.  Blocks in front of i are turned into pass-throughs and block i's input is replaced by the cached tensor, so the
model's own forward still does everything else (masks, rotary tables, final norm, head): the suffix executes the same
kernels on the same values as the full forward and the perplexities are bit-identical to the plain evaluator's.
A Linear that is called after the last block (lm_head, OPT project_out) is evaluated with every block skipped and its own
cached input substituted.  Average cost per evaluation: (N+1)/2N of a full forward, and ~0 for the head.

All candidate ratios of a layer in ONE suffix pass (`perplexities`, SURVEY 8f row 1, second half).  The factors of the sweep are
NESTED: rank-r factors are the first r columns / rows of the rank-r_max factors (each column of A = U sqrt(S) and row of B depends on
its own sigma only), so the rank-r output is a prefix sum over sigma-ordered components.  `MultiRankSVDLinear` holds the r_max factors
and gives batch element j the truncation r_j by zeroing the components >= r_j of z = B x before the A GEMM.  The cached hidden state
entering the layer's block is expanded to a batch of R identical copies, the suffix runs ONCE on [R, T, C] (masks and rotary tables
broadcast over the batch) and the logits give R perplexities.  GEMMs see R x T rows instead of T and the launch count drops R-fold.
Not bit-identical to the per-ratio path: the masked K = r_max GEMM and the batched suffix GEMMs accumulate in a different order
(fp16: ~1e-3 relative in the logits, ~1e-4 in the perplexity; fp32: 1e-6) — tolerance and the unchanged search trace are tested.
"""
import torch
import torch.nn as nn


def find_decoder_blocks(model):
    """The nn.ModuleList holding the repeated decoder blocks: the list whose members own the most nn.Linear parameters."""
    best, best_name, best_params = None, None, 0
    for name, mod in model.named_modules():
        if isinstance(mod, nn.ModuleList) and len(mod) > 0:
            n = sum(p.numel() for m in mod.modules() if isinstance(m, nn.Linear) for p in m.parameters(recurse=False))
            if n > best_params:
                best, best_name, best_params = mod, name, n
    return best_name, best


def _hidden_of(args, kwargs):
    return args[0] if len(args) > 0 else kwargs["hidden_states"]


def _with_hidden(args, kwargs, h):
    if len(args) > 0:
        return (h,) + tuple(args[1:]), kwargs
    kwargs = dict(kwargs)
    kwargs["hidden_states"] = h
    return args, kwargs


class MultiRankSVDLinear(nn.Module):
    """R nested truncations of one factorisation behind a single module: batch element j of the input gets rank ranks[j].
    A [out, r_max], B [r_max, in] are the fused, cast factors at the largest rank (what SVDLinear holds); forward(x) expects the
    batch dimension of x to be R (3-D [R, T, in], or 2-D [R*T, in] as OPT's flattened MLP inputs)."""

    def __init__(self, A, B, bias, ranks):
        super().__init__()
        self.A, self.B, self.bias = A, B, bias
        self.ranks = list(ranks)
        rmax = B.shape[0]
        idx = torch.arange(rmax, device=B.device)
        self.mask = torch.stack([(idx < r) for r in self.ranks])  # [R, r_max] bool
        self.group = 1          # calibration samples per pass: batch element b = j * group + s carries truncation j of sample s
        self._gmask = {1: self.mask}

    def forward(self, x):
        R, g = len(self.ranks), self.group
        flat = x.dim() == 2
        if flat:
            x = x.view(R * g, -1, x.shape[-1])
        assert x.shape[0] == R * g, f"MultiRankSVDLinear expects batch {R} x {g}, got {tuple(x.shape)}"
        mask = self._gmask.get(g)
        if mask is None:
            mask = self._gmask[g] = self.mask.repeat_interleave(g, dim=0)
        z = nn.functional.linear(x, self.B)                       # [R g, T, r_max]
        y = nn.functional.linear(torch.where(mask[:, None, :], z, torch.zeros((), dtype=z.dtype, device=z.device)), self.A, self.bias)  # select, not multiply: an inf in a masked component must not become NaN
        return y.view(-1, y.shape[-1]) if flat else y


class PrefixCachedEvaluator:
    def __init__(self, model, input_ids, limit, mem_fraction=0.5):
        self.model = model
        self.input_ids = input_ids
        nsamples, self.seqlen = input_ids.size()
        self.n = min(nsamples, limit)
        self.blocks_name, self.blocks = find_decoder_blocks(model)
        if self.blocks is None:
            raise RuntimeError("no decoder block list found")
        self.nblocks = len(self.blocks)
        self.block_index = {}  # Linear full name -> block index
        prefix = self.blocks_name + "."
        self.linear_names = {}
        for name, mod in model.named_modules():
            if isinstance(mod, nn.Linear):
                self.linear_names[mod] = name
                if name.startswith(prefix):
                    self.block_index[name] = int(name[len(prefix):].split(".")[0])
        self._capture(mem_fraction)

    # ------------------------------------------------------------------------------------------------------------
    def _budget_stride(self, per_block_bytes, mem_fraction):
        dev = self.model.device
        if dev.type != "cuda":
            return 1
        free, _ = torch.cuda.mem_get_info(dev)
        total = per_block_bytes * self.nblocks * self.n
        budget = free * mem_fraction
        stride = 1
        while total / stride > budget and stride < self.nblocks:
            stride += 1
        return stride

    @torch.no_grad()
    def _capture(self, mem_fraction):
        """One forward of the unmodified model per sample: block inputs, post-block Linear inputs, call order."""
        model = self.model
        self.cached = [dict() for _ in range(self.n)]  # sample -> {block idx: hidden states}
        self.tail_inputs = [dict() for _ in range(self.n)]  # sample -> {linear name: input tensor}
        self.tuple_out = {}
        order = []
        cur = {"i": 0}
        handles = []
        stride_box = {"s": None}

        def block_pre(idx):
            def hook(mod, args, kwargs):
                h = _hidden_of(args, kwargs)
                if stride_box["s"] is None:
                    stride_box["s"] = self._budget_stride(h.numel() * h.element_size(), mem_fraction)
                if idx % stride_box["s"] == 0:
                    self.cached[cur["i"]][idx] = h.detach().clone()
                order.append(("block", idx))
            return hook

        def block_post(idx):
            def hook(mod, args, out):
                self.tuple_out[idx] = isinstance(out, tuple)
            return hook

        for idx, blk in enumerate(self.blocks):
            handles.append(blk.register_forward_pre_hook(block_pre(idx), with_kwargs=True))
            handles.append(blk.register_forward_hook(block_post(idx)))

        outside = [(m, n) for m, n in self.linear_names.items() if n not in self.block_index]

        def lin_pre(name):
            def hook(mod, args):
                order.append(("linear", name))
                self.tail_inputs[cur["i"]][name] = args[0].detach().clone()
            return hook

        for m, n in outside:
            handles.append(m.register_forward_pre_hook(lin_pre(n)))
        try:
            for i in range(self.n):
                cur["i"] = i
                if i == 1:
                    order_first = list(order)
                ids = self.input_ids[i:i + 1, :-1].to(model.device)
                model(input_ids=ids, use_cache=False)
        finally:
            for h in handles:
                h.remove()
        self.stride = stride_box["s"] or 1
        first = order_first if self.n > 1 else order
        last_block_pos = max((p for p, (k, _) in enumerate(first) if k == "block"), default=-1)
        # Linears outside the blocks that run AFTER the last block can be evaluated from their own cached input
        self.after_blocks = {name for p, (k, name) in enumerate(first) if k == "linear" and p > last_block_pos}
        for i in range(self.n):
            self.tail_inputs[i] = {k: v for k, v in self.tail_inputs[i].items() if k in self.after_blocks}

    # ------------------------------------------------------------------------------------------------------------
    def _skip_blocks(self, upto):
        """Turn blocks [0, upto) into pass-throughs; returns the undo list."""
        undo = []
        for idx in range(upto):
            blk = self.blocks[idx]
            had = "forward" in blk.__dict__
            old = blk.__dict__.get("forward")
            tup = self.tuple_out.get(idx, False)

            def passthrough(*args, _tup=tup, **kwargs):
                h = _hidden_of(args, kwargs)
                return (h,) if _tup else h

            blk.forward = passthrough
            undo.append((blk, had, old))
        return undo

    @staticmethod
    def _restore(undo):
        for blk, had, old in undo:
            if had:
                blk.forward = old
            else:
                del blk.forward

    @torch.no_grad()
    def perplexity(self, full_name, current_module):
        """Calibration perplexity of the model as it is NOW (one Linear `full_name` swapped for `current_module`);
        same arithmetic as evaluate_perplexity on the part of the network behind the swap."""
        model = self.model
        cur = {"i": 0}
        undo, handle = [], None
        if full_name in self.block_index:
            start = (self.block_index[full_name] // self.stride) * self.stride
            if start > 0:
                undo = self._skip_blocks(start)

                def sub(mod, args, kwargs):
                    return _with_hidden(args, kwargs, self.cached[cur["i"]][start])

                handle = self.blocks[start].register_forward_pre_hook(sub, with_kwargs=True)
        elif full_name in self.after_blocks:
            undo = self._skip_blocks(self.nblocks)

            def sub(mod, args):
                return (self.tail_inputs[cur["i"]][full_name],) + tuple(args[1:])

            handle = current_module.register_forward_pre_hook(sub)
        # else: a Linear in front of the blocks — nothing can be reused, full forward
        nlls = []
        seqlen = self.seqlen
        try:
            for i in range(self.n):
                cur["i"] = i
                input_ids = self.input_ids[i:i + 1, :-1].to(model.device)
                labels = self.input_ids[i:i + 1, 1:].contiguous()
                logits = model(input_ids=input_ids, use_cache=False)[0]
                shift_labels = labels.to(model.device)
                loss = nn.CrossEntropyLoss()(logits.view(-1, logits.size(-1)), shift_labels.view(-1))
                nlls.append(loss.float() * seqlen)
        finally:
            if handle is not None:
                handle.remove()
            self._restore(undo)
        ppl = torch.exp(torch.stack(nlls).sum() / (len(nlls) * seqlen))
        return ppl.item()

    @torch.no_grad()
    def perplexities(self, full_name, multi_module, samples_per_pass=1):
        """Calibration perplexities of the R rank truncations held by `multi_module` (a MultiRankSVDLinear already installed in place
        of the Linear `full_name`): ONE suffix pass per `samples_per_pass` calibration samples on a batch of R x samples copies.  Returns a list of
        R floats, each with the arithmetic of evaluate_perplexity (mean over T-1 tokens times seqlen, evaluate_utils.py:95-114).
        Returns None when the layer sits in front of the decoder blocks (nothing to batch: use the per-ratio path).

        samples_per_pass > 1 (round 6): the cached block inputs of g samples are stacked, each repeated R times — batch element b = j g + s is
        truncation j of sample s — and the suffix runs once per g samples.  The forward is called with ONE sample's ids (everything in front of the
        substituted block is skipped, position ids and the causal mask are batch-1 and broadcast, exactly as for g = 1); every (j, s) gets its own
        CrossEntropyLoss against the labels of ITS sample, summed per truncation in sample order — the per-sample arithmetic, on GEMMs with g x
        the rows (tools/sweep_gemm_ceiling.py: a Llama-2-7B block costs 0.419 / 0.371 / 0.361 us per token at g = 1 / 2 / 4)."""
        model = self.model
        R = len(multi_module.ranks)
        g_max = max(1, min(int(samples_per_pass), self.n))
        cur = {"ids": [0]}

        def stacked(get):
            hs = [get(i) for i in cur["ids"]]
            if hs[0].dim() == 3:      # [1, T, C] per sample -> [R g, T, C], b = j g + s
                h = hs[0] if len(hs) == 1 else torch.cat(hs, 0)
                return h.expand(R, *h.shape[1:]) if len(hs) == 1 else h.repeat(R, 1, 1)
            h = hs[0] if len(hs) == 1 else torch.cat(hs, 0)       # OPT's flattened [T, C] inputs -> [R g T, C]
            return h.repeat(R, 1)

        undo, handle = [], None
        if full_name in self.block_index:
            start = (self.block_index[full_name] // self.stride) * self.stride
            undo = self._skip_blocks(start)

            def sub(mod, args, kwargs):
                if start not in self.cached[cur["ids"][0]]:   # (cannot happen: start is a multiple of the caching stride)
                    h = _hidden_of(args, kwargs)
                    return _with_hidden(args, kwargs, h.expand(R, *h.shape[1:]) if h.dim() == 3 else h.repeat(R, 1))
                return _with_hidden(args, kwargs, stacked(lambda i: self.cached[i][start]))

            handle = self.blocks[start].register_forward_pre_hook(sub, with_kwargs=True)
        elif full_name in self.after_blocks:
            undo = self._skip_blocks(self.nblocks)

            def sub(mod, args):
                return (stacked(lambda i: self.tail_inputs[i][full_name]),) + tuple(args[1:])

            handle = multi_module.register_forward_pre_hook(sub)
        else:
            return None
        nll = torch.zeros(R, dtype=torch.float32, device=model.device)
        seqlen = self.seqlen
        try:
            for i0 in range(0, self.n, g_max):
                ids = list(range(i0, min(self.n, i0 + g_max)))
                g = len(ids)
                cur["ids"] = ids
                multi_module.group = g
                input_ids = self.input_ids[i0:i0 + 1, :-1].to(model.device)
                logits = model(input_ids=input_ids, use_cache=False)[0]      # [R g, T-1, V]
                assert logits.shape[0] == R * g, f"batched suffix returned batch {logits.shape[0]}, expected {R} x {g}"
                for s, i in enumerate(ids):
                    labels = self.input_ids[i:i + 1, 1:].contiguous().to(model.device).view(-1)
                    for j in range(R):  # one CrossEntropyLoss per (truncation, sample), exactly the per-sample arithmetic
                        loss = nn.CrossEntropyLoss()(logits[j * g + s].view(-1, logits.size(-1)), labels)
                        nll[j] += loss.float() * seqlen
                del logits
        finally:
            multi_module.group = 1
            if handle is not None:
                handle.remove()
            self._restore(undo)
        return [torch.exp(nll[j] / (self.n * seqlen)).item() for j in range(R)]
