"""Llama-shaped toy module tree used by the host-logic and end-to-end tests (names matter, arithmetic is arbitrary)."""
import types

import torch
import torch.nn as nn


class Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(d, d, bias=False) for _ in range(4))


class Mlp(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = nn.Linear(d, f, bias=False), nn.Linear(d, f, bias=False), nn.Linear(f, d, bias=False)


class Layer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn, self.mlp = Attn(d), Mlp(d, f)

    def forward(self, h):
        a = self.self_attn
        h = h + a.o_proj(torch.tanh(a.q_proj(h)) * torch.sigmoid(a.k_proj(h)) + a.v_proj(h))
        m = self.mlp
        return h + m.down_proj(torch.nn.functional.silu(m.gate_proj(h)) * m.up_proj(h))


class TinyLM(nn.Module):
    def __init__(self, d=32, f=80, n_layers=2, vocab=50, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.config = types.SimpleNamespace(_name_or_path="golden/tiny_lm", vocab_size=vocab)
        self.model = nn.Module()
        self.model.embed_tokens = nn.Embedding(vocab, d)
        self.model.layers = nn.ModuleList([Layer(d, f) for _ in range(n_layers)])
        self.lm_head = nn.Linear(d, vocab, bias=False)

    @property
    def device(self):
        return self.lm_head.weight.device if isinstance(self.lm_head, nn.Linear) else next(self.parameters()).device

    def forward(self, input_ids=None, labels=None, **kw):
        h = self.model.embed_tokens(input_ids)
        for l in self.model.layers:
            h = l(h)
        logits = self.lm_head(h)
        if labels is not None:  # HF convention used by calib_fisher_info: (loss, logits)
            return (nn.functional.cross_entropy(logits.view(-1, logits.size(-1)), labels.reshape(-1)), logits)
        return (logits,)


def load_golden_tiny(golden):
    """TinyLM with the exact weights and scaling vectors the reference run used (tests/golden/tiny_lm_state.npz)."""
    st = golden.npz("tiny_lm_state.npz")
    model = TinyLM()
    sd = {k: torch.from_numpy(st[k]) for k in st.files if not k.startswith("scal::")}
    model.load_state_dict(sd)
    scal = {k[len("scal::"):]: torch.from_numpy(st[k]) for k in st.files if k.startswith("scal::")}
    return model, scal


def default_args(**kw):
    base = dict(scaling_method="abs_mean", alpha=0.5, n_calib_samples=3, calib_dataset="wikitext2", compress_kv_cache=False, rank_align=1,
                act_aware=True, sigma_fuse="UV", ppl_target=-1, param_ratio_target=0.8, kv_cache_ratio_target=-1)
    base.update(kw)
    return types.SimpleNamespace(**base)


class WideBlock(nn.Module):
    """one residual block around a 4096-wide square Linear (the BASELINE unit shape) between a tall and a wide projection"""

    def __init__(self, d, w):
        super().__init__()
        self.up_proj, self.mid_proj, self.down_proj = nn.Linear(d, w, bias=False), nn.Linear(w, w, bias=False), nn.Linear(w, d, bias=True)

    def forward(self, h):
        return h + self.down_proj(torch.tanh(self.mid_proj(torch.nn.functional.silu(self.up_proj(h)))))


class WideLM(nn.Module):
    """TinyLM-shaped module tree whose block holds ONE width x width Linear (width 4096 in the sharded GPU test: the two-level sweeps,
    the fused update + Gram kernel and the Cholesky-QR reduction all run) next to the small Linears of a TinyLM layer."""

    def __init__(self, d=64, width=4096, f=176, vocab=50, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.config = types.SimpleNamespace(_name_or_path="golden/wide_lm", vocab_size=vocab)
        self.model = nn.Module()
        self.model.embed_tokens = nn.Embedding(vocab, d)
        self.model.layers = nn.ModuleList([Layer(d, f), WideBlock(d, width)])
        self.lm_head = nn.Linear(d, vocab, bias=False)
        with torch.no_grad():
            self.model.layers[1].mid_proj.weight.mul_(0.5)

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, input_ids=None, labels=None, **kw):
        h = self.model.embed_tokens(input_ids)
        for l in self.model.layers:
            h = l(h)
        logits = self.lm_head(h)
        if labels is not None:
            return (nn.functional.cross_entropy(logits.view(-1, logits.size(-1)), labels.reshape(-1)), logits)
        return (logits,)


# ---- stand-ins shared by the host-logic tests ------------------------------------------------------------------------
class OracleTwoLinear(nn.Module):
    """what SVDLinear is to a forward pass, built from the CPU oracle's factors (tests only: the product has no CPU path)"""

    def __init__(self, A, B, rank):
        super().__init__()
        self.A, self.B, self.truncation_rank = A, B, rank

    def forward(self, x):
        return torch.nn.functional.linear(torch.nn.functional.linear(x, self.B), self.A)


def oracle_from_linear(linear, param_ratio, act_aware=False, ic_split=1, oc_split=1, alpha=1, sigma_fuse="UV", rank_align=1):
    from oracle import asvd_oracle as O
    o = O.from_linear_oracle(linear.weight.data, getattr(linear, "scaling_diag_matrix", None), param_ratio, alpha=alpha, act_aware=act_aware,
                             sigma_fuse=sigma_fuse, rank_align=rank_align)
    return OracleTwoLinear(o["A"], o["B"], o["rank"])


def parse_search_trace(lines):
    """[(low, mid, high, value, ratio)] of the `low=.. mid=.., high=.., ppl=.., param_ratio=..` / `now_ratio=.., params=..` lines"""
    import re
    out = []
    for l in lines:
        m = re.match(r"low=(\d+) mid=(\d+), high=(\d+), (?:ppl|now_ratio)=([^,]+), (?:param_ratio=(.+)|params=\((.+)\))$", l)
        if m:
            out.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), m.group(5) or m.group(6)))
    return out


def proxy_linear(out_f, in_f):
    """an nn.Linear of the given shape whose weight is ONE float expanded to [out, in]: numel / shape as the real layer, 4 bytes of storage"""
    lin = nn.Linear(1, 1, bias=False)
    lin.in_features, lin.out_features = in_f, out_f
    lin.weight = nn.Parameter(torch.zeros(1, 1).expand(out_f, in_f), requires_grad=False)
    return lin


class ShapedLlama(nn.Module):
    """the module tree (names, order, weight shapes) of a Llama checkpoint without its memory: search / sharding logic at model scale"""

    def __init__(self, hidden=4096, inter=11008, layers=32, vocab=32000):
        super().__init__()
        self.model = nn.Module()
        blocks = []
        for _ in range(layers):
            blk = nn.Module()
            blk.self_attn = nn.Module()
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(blk.self_attn, n, proxy_linear(hidden, hidden))
            blk.mlp = nn.Module()
            blk.mlp.gate_proj, blk.mlp.up_proj, blk.mlp.down_proj = proxy_linear(inter, hidden), proxy_linear(inter, hidden), proxy_linear(hidden, inter)
            blocks.append(blk)
        self.model.layers = nn.ModuleList(blocks)
        self.lm_head = proxy_linear(vocab, hidden)
