"""Prefix-cached sweep evaluator (asvd4llm_amd/sweep_eval.py) == plain evaluate_perplexity, bit for bit, on CPU.
The swapped module is an arbitrary perturbed nn.Linear, so no GPU kernel is involved."""
import copy

import pytest
import torch
import torch.nn as nn

from asvd4llm_amd.evaluate_utils import evaluate_perplexity
from asvd4llm_amd.model_zoo import random_init_model
from asvd4llm_amd.sensitivity import collect_linear_info
from asvd4llm_amd.sweep_eval import PrefixCachedEvaluator, find_decoder_blocks
from tests.tiny_lm import TinyLM


def _perturbed(linear, seed):
    g = torch.Generator().manual_seed(seed)
    new = copy.deepcopy(linear)
    new.weight.data += 0.05 * new.weight.data.std() * torch.randn(new.weight.shape, generator=g)
    return new


def _check_model(model, vocab, n=3, T=24):
    torch.manual_seed(5)
    ids = torch.randint(0, vocab, (n, T))
    model.eval()
    ev = PrefixCachedEvaluator(model, ids, n)
    info = collect_linear_info(model)
    base = evaluate_perplexity(model, ids, n)
    checked = set()
    for k, (lin, meta) in enumerate(info.items()):
        swapped = _perturbed(lin, k)
        setattr(meta["father"], meta["name"], swapped)
        try:
            plain = evaluate_perplexity(model, ids, n)
            fused = ev.perplexity(meta["full_name"], swapped)
        finally:
            setattr(meta["father"], meta["name"], lin)
        assert plain == fused, (meta["full_name"], plain, fused)
        assert plain != base
        checked.add("block" if meta["full_name"] in ev.block_index else "tail" if meta["full_name"] in ev.after_blocks else "full")
    assert evaluate_perplexity(model, ids, n) == base  # nothing left patched
    return ev, checked


def test_blocks_found_and_values_identical_tiny_lm():
    model = TinyLM(n_layers=3)
    name, blocks = find_decoder_blocks(model)
    assert name == "model.layers" and len(blocks) == 3
    ev, kinds = _check_model(model, 50)
    assert kinds == {"block", "tail"} and ev.after_blocks == {"lm_head"}


@pytest.mark.parametrize("name", ["tiny-llama", "tiny-opt"])
def test_values_identical_hf(name):
    model = random_init_model(name, dtype=torch.float32, seed=1)
    ev, kinds = _check_model(model, model.config.vocab_size)
    assert "block" in kinds and "tail" in kinds
    assert ev.nblocks == model.config.num_hidden_layers


def test_stride_fallback_identical():
    model = TinyLM(n_layers=4)
    torch.manual_seed(1)
    ids = torch.randint(0, 50, (2, 16))
    ev = PrefixCachedEvaluator(model, ids, 2)
    # emulate a memory-limited capture: keep only every 2nd block input
    ev.stride = 2
    for c in ev.cached:
        for k in [k for k in c if k % 2]:
            del c[k]
    lin = model.model.layers[3].mlp.up_proj
    swapped = _perturbed(lin, 0)
    model.model.layers[3].mlp.up_proj = swapped
    assert ev.perplexity("model.layers.3.mlp.up_proj", swapped) == evaluate_perplexity(model, ids, 2)


# ---------------------------------------------------------------------------------------------------------------------
# all candidate ranks of a layer in ONE batched suffix pass (MultiRankSVDLinear / PrefixCachedEvaluator.perplexities)
def _factors(linear, rmax, seed=0):
    """exact fp64 SVD factors at rank rmax, fused 'UV' like SVDLinear (CPU stand-in for the device factorisation)"""
    U, S, Vh = torch.linalg.svd(linear.weight.data.double(), full_matrices=False)
    A = (U[:, :rmax] * S[:rmax].sqrt()).to(linear.weight.dtype)
    B = (S[:rmax].sqrt()[:, None] * Vh[:rmax]).to(linear.weight.dtype)
    return A, B


class _TwoLin(nn.Module):
    def __init__(self, A, B, bias):
        super().__init__()
        self.A, self.B, self.bias = A, B, bias

    def forward(self, x):
        return nn.functional.linear(nn.functional.linear(x, self.B), self.A, self.bias)


def _check_multi(model, vocab, n=3, T=20, tol=2e-6):
    from asvd4llm_amd.sweep_eval import MultiRankSVDLinear
    torch.manual_seed(7)
    ids = torch.randint(0, vocab, (n, T))
    model.eval()
    ev = PrefixCachedEvaluator(model, ids, n)
    info = collect_linear_info(model)
    kinds = set()
    for lin, meta in info.items():
        name = meta["full_name"]
        kmax = min(lin.in_features, lin.out_features)
        ranks = sorted({max(1, kmax // 4), max(1, kmax // 2), max(1, (3 * kmax) // 4)})
        A, B = _factors(lin, max(ranks))
        bias = lin.bias.data if lin.bias is not None else None
        multi = MultiRankSVDLinear(A, B, bias, ranks)
        setattr(meta["father"], meta["name"], multi)
        try:
            got = ev.perplexities(name, multi)
            # several calibration samples per suffix pass (round 6): 2 (an uneven last chunk: n = 3) and all 3 at once — the same per-sample arithmetic
            for g_ in (2, 3):
                got_g = ev.perplexities(name, multi, samples_per_pass=g_)
                assert (got is None) == (got_g is None)
                if got is not None:
                    for a_, b_ in zip(got, got_g):
                        assert abs(a_ - b_) <= tol * abs(a_), (name, g_, a_, b_)
            assert multi.group == 1
        finally:
            setattr(meta["father"], meta["name"], lin)
        if got is None:
            kinds.add("front")
            continue
        kinds.add("block" if name in ev.block_index else "tail")
        for r, g in zip(ranks, got):
            two = _TwoLin(A[:, :r].contiguous(), B[:r].contiguous(), bias)  # the per-ratio module: the first r components
            setattr(meta["father"], meta["name"], two)
            try:
                ref = ev.perplexity(name, two)
            finally:
                setattr(meta["father"], meta["name"], lin)
            assert abs(g - ref) <= tol * abs(ref), (name, r, g, ref)
    return kinds


def test_multi_rank_pass_matches_per_ratio_tiny_lm():
    assert _check_multi(TinyLM(n_layers=3), 50) == {"block", "tail"}


@pytest.mark.parametrize("name", ["tiny-llama", "tiny-opt"])
def test_multi_rank_pass_matches_per_ratio_hf(name):
    """HF Llama / OPT: masks and rotary tables of the batch-1 input broadcast over the R-copy batch substituted at the block input;
    OPT's MLP Linears receive the flattened [R*T, C] view"""
    model = random_init_model(name, dtype=torch.float32, seed=2)
    kinds = _check_multi(model, model.config.vocab_size, tol=5e-6)
    assert "block" in kinds and "tail" in kinds
