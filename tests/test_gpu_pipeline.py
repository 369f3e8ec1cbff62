"""GPU end-to-end parity: SVDLinear.from_linear against the reference-generated fixtures, and the full pipeline
(hook -> sweep -> search -> decomposition) on the toy LM against the values the reference produced for the same weights."""
import contextlib
import io
import numpy as np
import os

import pytest
import torch
import torch.nn as nn

from oracle import asvd_oracle as O
from tests.tiny_lm import default_args, load_golden_tiny

pytestmark = pytest.mark.gpu


def test_from_linear_vs_reference_fixtures(gpu, golden):
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    meta = golden.json("svd_linear_meta.json")
    g = golden.npz("svd_linear.npz")
    lins = {}
    for rec in meta:
        ci = rec["case"]
        dt = torch.float16 if rec["dtype"] == "float16" else torch.float32
        if ci not in lins:
            W = torch.from_numpy(g[f"c{ci}_W"])
            lin = nn.Linear(rec["in"], rec["out"], bias=rec["has_bias"]).to(dt)
            lin.weight.data = W
            lin = lin.to(gpu)
            lin.scaling_diag_matrix = torch.from_numpy(g[f"c{ci}_scal"]).to(gpu)
            lins[ci] = (lin, W)
        lin, W = lins[ci]
        m = SVDLinear.from_linear(lin, rec["ratio"], act_aware=True, alpha=rec["alpha"], sigma_fuse=rec["fuse"])
        assert isinstance(m, SVDLinear) and m.truncation_rank == rec["rank"]
        A, B = m.ALinear.weight.data, m.BLinear.weight.data
        A_ref, B_ref = torch.from_numpy(g[rec["key"] + "_A"]), torch.from_numpy(g[rec["key"] + "_B"])
        assert A.dtype == dt and A.shape == A_ref.shape and B.shape == B_ref.shape and A.is_contiguous() and B.is_contiguous()
        assert (m.ALinear.bias is not None) == rec["has_bias"]
        P_ref = A_ref.double() @ B_ref.double()
        tol = 3e-3 if dt == torch.float16 else 1e-3  # BASELINE contract: <= 1e-3 |W|_F (fp16 factors add their rounding)
        s_ref = O.make_scale(torch.from_numpy(g[f"c{ci}_scal"]), rec["alpha"])
        e_live, e_scaled = O.recon_parity(A, B, A_ref, B_ref, W, s_ref)
        assert e_live <= tol and e_scaled <= tol, (rec["key"], e_live, e_scaled)
        if rec["fuse"] == "U":
            sv, sv_ref = A.double().cpu().norm(dim=0), A_ref.double().norm(dim=0)
            assert ((sv - sv_ref).abs() / sv_ref).max().item() <= (2e-3 if dt == torch.float16 else 1e-4)
        # forward parity of the swapped-in module
        x = torch.randn(5, rec["in"], generator=torch.Generator().manual_seed(0)).to(dt).to(gpu)
        y = m(x).float().cpu()
        live = O.live_channels(s_ref)
        xl = x.cpu().double() * live.double().view(1, -1)  # dead input channels carry the reference's amplified noise
        y = m((x * live.to(gpu).to(dt).view(1, -1))).float().cpu()
        y_ref = (xl @ P_ref.T + (lin.bias.data.double().cpu() if rec["has_bias"] else 0)).float()
        assert (y - y_ref).norm() / (y_ref.norm() + 1e-9) <= (2e-2 if dt == torch.float16 else 1e-3)


def test_svd_is_cached_once_per_layer(gpu):
    from asvd4llm_amd import ops
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    lin = nn.Linear(96, 128, bias=True).half().to(gpu)
    lin.scaling_diag_matrix = torch.rand(96, device=gpu).half()
    calls = {"n": 0}
    real = ops.svd

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    ops.svd = counting
    try:
        ranks = []
        for ratio in (0.4, 0.5, 0.6, 0.7, 0.8, 0.9):
            m = SVDLinear.from_linear(lin, ratio, act_aware=True, alpha=0.5)
            ranks.append(m.truncation_rank)
        assert calls["n"] == 1 and ranks == sorted(ranks)
        lin.weight.data.add_(1.0)  # in-place change bumps the version -> re-factorised
        SVDLinear.from_linear(lin, 0.5, act_aware=True, alpha=0.5)
        assert calls["n"] == 2
        # edits that KEEP the sum of squares: a sign flip of one row, a permutation of two rows, a changed statistic
        lin.weight.data[3].neg_()
        SVDLinear.from_linear(lin, 0.5, act_aware=True, alpha=0.5)
        assert calls["n"] == 3
        lin.weight.data[[0, 1]] = lin.weight.data[[1, 0]]
        SVDLinear.from_linear(lin, 0.5, act_aware=True, alpha=0.5)
        assert calls["n"] == 4
        lin.scaling_diag_matrix[[0, 1]] = lin.scaling_diag_matrix[[1, 0]]
        SVDLinear.from_linear(lin, 0.5, act_aware=True, alpha=0.5)
        assert calls["n"] == 5
        SVDLinear.from_linear(lin, 0.6, act_aware=True, alpha=0.5)  # untouched: cached
        assert calls["n"] == 5
    finally:
        ops.svd = real


def test_fallback_behaviour_matches_reference(gpu, capsys):
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    os.environ["ASVD_STRICT"] = "0"
    try:
        lin = nn.Linear(64, 64).to(gpu)
        lin.weight.data[0, 0] = float("nan")
        m = SVDLinear.from_linear(lin, 0.5)
        out = capsys.readouterr().out
        assert isinstance(m, nn.Linear) and not isinstance(m, SVDLinear) and ("svd failed" in out or "nan in" in out)
        assert m.weight.device.type == "cuda" and m.in_features == 64
    finally:
        os.environ["ASVD_STRICT"] = "1"
    with pytest.raises(Exception):
        SVDLinear.from_linear(lin, 0.5)


def test_full_pipeline_vs_reference_run(gpu, golden, tmp_path, monkeypatch):
    """hook -> ppl sweep -> binary search -> decomposition on the toy LM; numbers from the reference run on identical weights
    (with its SVD patched to the exact one) are the expected values."""
    from asvd4llm_amd.act_aware_utils import calib_input_distribution
    from asvd4llm_amd.binary_search import binary_search_truncation_rank
    from asvd4llm_amd.evaluate_utils import evaluate_perplexity
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    from asvd4llm_amd.sensitivity import calib_sensitivity_ppl, calib_sensitivity_stable_rank
    monkeypatch.chdir(tmp_path)
    t = golden.json("tiny_lm.json")
    model, scal_ref = load_golden_tiny(golden)
    model = model.to(gpu)
    calib = [{"input_ids": torch.tensor(ids)} for ids in t["calib_ids"]]
    args = default_args()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        calib_input_distribution(model, calib, "abs_mean", use_cache=False)
    assert os.path.exists("cache/golden_tiny_lm_calib_input_distribution_abs_mean.pt")
    for n, mod in model.named_modules():
        if isinstance(mod, nn.Linear):
            got, want = mod.scaling_diag_matrix.float().cpu(), scal_ref[n].float()
            assert got.shape == want.shape and ((got - want).abs() <= 2e-6 * want.abs() + 1e-7).all(), n
    cached = torch.load("cache/golden_tiny_lm_calib_input_distribution_abs_mean.pt", map_location="cpu")
    assert list(cached.keys()) == [n for n, m in model.named_modules() if isinstance(m, nn.Linear)]

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        sens = calib_sensitivity_ppl(model, calib, args, use_cache=False)
    assert list(sens.keys()) == t["order"]
    for name, d in t["sensitivity_ppl"].items():
        assert list(sens[name].keys()) == [float(r) for r in d.keys()]
        for r, v in d.items():
            assert abs(sens[name][float(r)] - v) <= 2e-4 * v, (name, r, sens[name][float(r)], v)
    # every layer restored to the raw Linear after its sweep
    assert not any(isinstance(m, SVDLinear) for m in model.modules())

    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        sr = calib_sensitivity_stable_rank(model, calib, args, use_cache=False)
    for name, d in t["sensitivity_stable_rank"].items():
        for r, v in d.items():
            assert abs(float(sr[name][float(r)]) - v) <= 1e-4 * abs(v)

    rec = t["search"]["ratio0.8"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        binary_search_truncation_rank(model, {k: {float(r): v for r, v in d.items()} for k, d in rec["sens"].items()}, calib, args)
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("low=") or l.startswith("===")]
    assert lines == rec["trace"]
    got = {n: (m.truncation_rank if isinstance(m, SVDLinear) else -1) for n, m in model.named_modules()
           if isinstance(m, SVDLinear) or (isinstance(m, nn.Linear) and not n.endswith("ALinear") and not n.endswith("BLinear"))}
    assert got == rec["ranks"]
    ids = torch.cat([c["input_ids"] for c in calib], 0)
    ppl = evaluate_perplexity(model, ids, 3)
    assert abs(ppl - rec["ppl_after"]) <= 2e-4 * rec["ppl_after"]
    sd = model.state_dict()
    assert any(k.endswith("ALinear.weight") for k in sd) and any(k.endswith("BLinear.weight") for k in sd)


@pytest.mark.parametrize("tag", ["ppl54.1", "ppl54.3"])
def test_ppl_target_search_vs_reference_run(gpu, golden, tag):
    """--ppl_target with the HIP path: every probe factorises the whole toy LM (from the cached exact SVD of each layer) and measures its
    perplexity; bisection steps, perplexities (2e-4) and final ranks are the reference's run on the same weights (binary_search.py:64-87)."""
    from asvd4llm_amd.binary_search import binary_search_truncation_rank
    from asvd4llm_amd.evaluate_utils import evaluate_perplexity
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    from tests.tiny_lm import parse_search_trace
    t = golden.json("tiny_lm.json")
    rec = golden.json("search_extra.json")["tiny_ppl_target"][tag]
    model, scal = load_golden_tiny(golden)
    model = model.to(gpu)
    for n, m in model.named_modules():
        if isinstance(m, nn.Linear):
            m.scaling_diag_matrix = scal[n].to(gpu)
    calib = [{"input_ids": torch.tensor(ids)} for ids in t["calib_ids"]]
    sens = {k: {float(r): v for r, v in d.items()} for k, d in t["sensitivity_ppl"].items()}
    args = default_args(ppl_target=rec["ppl_target"], param_ratio_target=-1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        binary_search_truncation_rank(model, sens, calib, args)
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("low=") or l.startswith("===")]
    got, want = parse_search_trace(lines), parse_search_trace(rec["trace"])
    assert len(got) == len(want) >= 6
    for g, w in zip(got, want):
        assert g[:3] == w[:3] and abs(g[3] - w[3]) <= 2e-4 * w[3] and g[4] == w[4], (g, w)
    ranks = {n: (m.truncation_rank if isinstance(m, SVDLinear) else -1) for n, m in model.named_modules()
             if isinstance(m, SVDLinear) or (isinstance(m, nn.Linear) and not n.endswith("ALinear") and not n.endswith("BLinear"))}
    assert ranks == rec["ranks"]
    ids = torch.cat([c["input_ids"] for c in calib], 0)
    assert abs(evaluate_perplexity(model, ids, 3) - rec["ppl_after"]) <= 2e-4 * rec["ppl_after"]


def test_prefactorize_batches_same_shape_layers(gpu):
    """model-level batching: same-shape Linears are factorised concurrently and land in the cache from_linear uses"""
    from asvd4llm_amd import ops
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    torch.manual_seed(0)
    lins = []
    for i in range(5):
        l = nn.Linear(96, 160, bias=False).half().to(gpu)
        l.scaling_diag_matrix = (torch.rand(96, device=gpu) * 4).half()
        lins.append(l)
    odd = nn.Linear(160, 96, bias=False).half().to(gpu)
    odd.scaling_diag_matrix = (torch.rand(160, device=gpu) * 4).half()
    lins.append(odd)
    calls = []
    real = ops.svd_batched

    def counting(mats, *a, **k):
        calls.append(len(mats))
        return real(mats, *a, **k)

    ops.svd_batched = counting
    try:
        ranks = {l: SVDLinear.compute_rank(l, 0.9) for l in lins}
        SVDLinear.prefactorize(lins, act_aware=True, alpha=0.5, ranks=ranks, max_batch=1)   # problems below 1024 columns: chunks of 4 * max_batch
        assert sorted(calls) == [1, 1, 4]
        n = len(calls)
        for l in lins:
            m = SVDLinear.from_linear(l, 0.6, act_aware=True, alpha=0.5)  # smaller rank: a slice of the cached factors
            assert isinstance(m, SVDLinear)
            o = O.from_linear_oracle(l.weight.data.cpu(), l.scaling_diag_matrix.cpu(), 0.6, alpha=0.5, act_aware=True)
            e_live, e_scaled = O.recon_parity(m.ALinear.weight.data, m.BLinear.weight.data, o["A"], o["B"], l.weight.data.cpu(), o["s"])
            assert e_live <= 3e-3 and e_scaled <= 3e-3
        assert len(calls) == n  # no re-factorisation
    finally:
        ops.svd_batched = real


def test_topk_request_converges_leading_part(gpu):
    from asvd4llm_amd import ops
    from tests.test_gpu_svd import llm_like
    W, s = llm_like(768, 768)
    So = torch.linalg.svdvals(O.scaled_weight(W, s))
    U, S, V, info = ops.svd(W.to(gpu), s.to(gpu), k=200)
    Uf, Sf, Vf, info_f = ops.svd(W.to(gpu), s.to(gpu))
    assert info.status == 0 and info.sweeps <= info_f.sweeps
    assert O.sigma_rel_err(S.cpu(), So, 200) <= 1e-4
    R = (U.double() * S.double()) @ V.double().T
    Rf = (Uf[:, :200].double() * Sf[:200].double()) @ Vf[:, :200].double().T
    assert ((R - Rf).norm() / Rf.norm()).item() <= 1e-4


def test_fisher_calibration_vs_cpu_restatement(gpu, tmp_path, monkeypatch):
    """--scaling_method fisher*: calib_fisher_info (act_aware_utils.py:8-44) on a tiny HF Llama, GPU kernel statistic vs the same
    torch-CPU ops the reference runs, then a fisher+abs_mean decomposition through from_linear."""
    import copy
    from asvd4llm_amd.act_aware_utils import calib_fisher_info
    from asvd4llm_amd.model_zoo import random_init_model
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    monkeypatch.chdir(tmp_path)
    cpu_model = random_init_model("tiny-llama", dtype=torch.float32, seed=1)
    gpu_model = copy.deepcopy(cpu_model).to(gpu)
    g = torch.Generator().manual_seed(3)
    calib = [{"input_ids": torch.randint(0, 512, (1, 33), generator=g)} for _ in range(2)]
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        calib_fisher_info(gpu_model, calib, use_cache=False)
    assert os.path.exists("cache/tiny-llama_calib_fisher_info.pt")
    # CPU restatement with the oracle's update rule
    acc = {}
    for batch in calib:
        out = cpu_model(input_ids=batch["input_ids"][:, :-1], labels=batch["input_ids"][:, 1:])
        out[0].backward()
        for n, m in cpu_model.named_modules():
            if isinstance(m, nn.Linear):
                acc[n] = O.fisher_update(acc.get(n), m.weight.grad)
        cpu_model.zero_grad()
    for n, m in gpu_model.named_modules():
        if isinstance(m, nn.Linear):
            want = (acc[n] / len(calib)).sqrt()
            got = m.fisher_info.float().cpu()
            assert got.shape == want.shape
            assert ((got - want).abs() <= 2e-3 * want.abs() + 1e-9).all(), n
    lin = gpu_model.model.layers[0].mlp.down_proj
    lin.scaling_diag_matrix = torch.rand(lin.in_features, device=gpu) + 0.5
    m = SVDLinear.from_linear(lin, 0.5, act_aware=True, alpha=0.5)
    o = O.from_linear_oracle(lin.weight.data.cpu(), lin.scaling_diag_matrix.cpu(), 0.5, alpha=0.5, act_aware=True, fisher_info=lin.fisher_info.cpu())
    e_live, e_scaled = O.recon_parity(m.ALinear.weight.data, m.BLinear.weight.data, o["A"], o["B"], lin.weight.data.cpu(), o["s"])
    assert e_live <= 1e-3 and e_scaled <= 1e-3


def test_kv_cache_mode_and_rank_align(gpu, golden, tmp_path, monkeypatch):
    """KV-cache compression mode: 19 candidate ratios up to 1.9 per layer from ONE factorisation, search restricted to
    k_proj / v_proj with default ratio 2, plus rank_align."""
    from asvd4llm_amd import ops
    from asvd4llm_amd.binary_search import binary_search_truncation_rank
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    from asvd4llm_amd.sensitivity import calib_sensitivity_ppl
    monkeypatch.chdir(tmp_path)
    t = golden.json("tiny_lm.json")
    model, scal = load_golden_tiny(golden)
    model = model.to(gpu)
    for n, m in model.named_modules():
        if isinstance(m, nn.Linear):
            m.scaling_diag_matrix = scal[n].to(gpu)
    calib = [{"input_ids": torch.tensor(ids)} for ids in t["calib_ids"]]
    args = default_args(compress_kv_cache=True, kv_cache_ratio_target=0.5, rank_align=4)
    calls = {"n": 0}
    real = ops.svd_batched

    def counting(mats, *a, **k):
        calls["n"] += len(mats)
        return real(mats, *a, **k)

    ops.svd_batched = counting
    # the reference sweeps EVERY Linear with ratios up to 1.9 in this mode; for the non-square ones the rank then exceeds min(in, out),
    # torch.svd_lowrank raises and the reference substitutes a random Linear ("svd failed ...") - reproduced in non-strict mode
    os.environ["ASVD_STRICT"] = "0"
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            sens = calib_sensitivity_ppl(model, calib, args, use_cache=False)
    finally:
        ops.svd_batched = real
        os.environ["ASVD_STRICT"] = "1"
    assert "svd failed" in buf.getvalue()
    assert calls["n"] == 15  # one factorisation per Linear for all 19 ratios
    assert all(len(v) == 19 for v in sens.values()) and list(sens.keys()) == t["order"]
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        binary_search_truncation_rank(model, sens, calib, args)
    for n, m in model.named_modules():
        if isinstance(m, SVDLinear):
            assert ("k_proj" in n or "v_proj" in n) and m.truncation_rank % 4 == 0
            ratio = model._asvd_layers_min_ratio[n]
            assert m.truncation_rank == O.rank_from_ratio(32, 32, ratio, 4)
    assert any(isinstance(m, SVDLinear) for m in model.modules())
    assert all(v == 2 for k, v in model._asvd_layers_min_ratio.items() if isinstance(dict(model.named_modules())[k], nn.Linear))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny-llama", "tiny-opt"])
def test_fused_sweep_equals_plain_sweep(gpu, name, tmp_path, monkeypatch):
    """the prefix-cached evaluator returns exactly the perplexities of full forwards (fp16 HF model on the GPU)."""
    from asvd4llm_amd.act_aware_utils import calib_input_distribution
    from asvd4llm_amd.datautils import get_calib_data
    from asvd4llm_amd.model_zoo import random_init_model
    from asvd4llm_amd.sensitivity import calib_sensitivity_ppl
    monkeypatch.chdir(tmp_path)
    model = random_init_model(name, dtype=torch.float16, seed=3).to(gpu)
    calib = get_calib_data("synthetic", None, name, 2, seed=7, vocab_size=model.config.vocab_size, seqlen=128)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        calib_input_distribution(model, calib, "abs_mean", False)
        fused = calib_sensitivity_ppl(model, calib, default_args(n_calib_samples=2, calib_dataset="synthetic", fused_sweep=True, sweep_samples_per_pass=1), use_cache=False)
        plain = calib_sensitivity_ppl(model, calib, default_args(n_calib_samples=2, calib_dataset="synthetic", fused_sweep=False), use_cache=False)
        # round 6: several calibration samples per batched suffix pass (opt-in, --sweep_samples_per_pass; here both samples in one pass)
        batched = calib_sensitivity_ppl(model, calib, default_args(n_calib_samples=2, calib_dataset="synthetic", fused_sweep=True, sweep_samples_per_pass=4), use_cache=False)
    assert list(fused.keys()) == list(plain.keys()) == list(batched.keys())
    assert fused == plain
    # the same per-sample arithmetic on GEMMs with twice the rows: fp16 accumulation order may differ -> equal to 2e-4 relative (VERDICT r5 task 6)
    for name_, d in plain.items():
        for ratio, v in d.items():
            assert abs(batched[name_][ratio] - v) <= 2e-4 * abs(v), (name_, ratio, batched[name_][ratio], v)


def test_from_linear_vs_reference_mid_size_fixtures(gpu, golden):
    """opt-125m shapes (BASELINE configs[0]) against what the imported reference produced for the same seeded inputs:
    sigma over the retained rank <= 1e-4, the compressed layer applied to 16 probe vectors <= 1e-3 |W x|."""
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    meta = golden.json("svd_mid_meta.json")
    g = golden.npz("svd_mid.npz")
    for rec in meta:
        ci = rec["case"]
        W, scal = O.synth_linear_numpy(rec["out"], rec["in"], rec["seed"])
        assert O.tensor_checksum(W, scal) == rec["inputs_sha256"]
        lin = nn.Linear(rec["in"], rec["out"], bias=False).to(torch.float16)
        lin.weight.data = W
        lin = lin.to(gpu)
        lin.scaling_diag_matrix = scal.to(gpu)
        m = SVDLinear.from_linear(lin, rec["ratio"], act_aware=True, alpha=rec["alpha"], sigma_fuse="UV")
        r = rec["rank"]
        assert isinstance(m, SVDLinear) and m.truncation_rank == r
        _, S, _, _ = SVDLinear.factorize(lin, act_aware=True, alpha=rec["alpha"], k=r)
        S_ref = torch.from_numpy(g[f"m{ci}_sigma"])
        assert O.sigma_rel_err(S.cpu(), S_ref, r) <= 1e-4
        X = torch.from_numpy(g[f"m{ci}_probe_x"])
        y = (m.ALinear.weight.data.float() @ (m.BLinear.weight.data.float() @ X.to(gpu))).cpu()
        y_ref, wx = torch.from_numpy(g[f"m{ci}_probe_y"]), torch.from_numpy(g[f"m{ci}_probe_wx"])
        assert ((y - y_ref).norm() / wx.norm()).item() <= 1e-3, (ci, ((y - y_ref).norm() / wx.norm()).item())


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_world1_allgather_and_exchange():
    """The nccl (= RCCL) branch on the MI355X: a world-size-1 process group pushes the sensitivity all-gather
    (all_gather_into_tensor on a CUDA fp64 buffer) and the factor broadcast through RCCL."""
    import torch.distributed as dist
    from asvd4llm_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29000 + os.getpid() % 2000)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        names = ["a", "b", "c"]
        ratios = [0.4, 0.9]
        local = {"a": {0.4: 1.25, 0.9: float("nan")}, "b": {0.4: float("inf"), 0.9: 3.000000123}, "c": {0.4: -1.0, 0.9: 0.0}}
        full = parallel.allgather_sensitivities(local, names, ratios, [0, 0, 0])
        assert list(full) == names
        for n in names:
            for r in ratios:
                a, b = full[n][r], local[n][r]
                assert (a != a and b != b) or a == b
        # broadcast path of exchange_factors with the only rank as owner: nothing to receive, but the RCCL broadcasts run
        lin = torch.nn.Linear(64, 48, bias=True).half().cuda()
        from asvd4llm_amd.modules.svd_linear import SVDLinear
        father = torch.nn.Module()
        father.x = SVDLinear._from_factors(torch.randn(48, 8, device="cuda").half(), torch.randn(8, 64, device="cuda").half(), lin.bias.data, 8)
        got = parallel.exchange_factors([("x", father, "x", lin)], {"x": 0}, mode="all")
        torch.cuda.synchronize()
        assert got == 0 and isinstance(father.x, SVDLinear)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_subgroup_under_a_gloo_default_group():
    """bench.py's arrangement at N > 1 (round 6): the DEFAULT group is gloo (timing barriers, MAX-reduce, bookkeeping on host tensors) and the path's
    collectives run on a separate RCCL group that parallel.GROUP points at.  World size 1 on the one GPU there is: host collectives on the default
    group, the probe all-reduce + the sensitivity all-gather + the factor broadcast on the RCCL group, then back to the default group."""
    import torch.distributed as dist
    from asvd4llm_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(31000 + os.getpid() % 2000)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert parallel.backend() == "gloo" and parallel._comm_device().type == "cpu"
        t = torch.tensor([3.5], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the timing reduction of bench.py: a host tensor on the default group
        dist.barrier()
        grp = dist.new_group(backend="nccl")
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe, group=grp)
        torch.cuda.synchronize()
        assert int(probe.item()) == 1
        parallel.set_group(grp)
        assert parallel.backend() == "nccl" and parallel._comm_device().type == "cuda" and parallel.world() == (0, 1)
        names, ratios = ["a", "b"], [0.4, 0.9]
        local = {"a": {0.4: 1.25, 0.9: float("nan")}, "b": {0.4: float("inf"), 0.9: 3.000000123}}
        full = parallel.allgather_sensitivities(local, names, ratios, [0, 0])
        for n in names:
            for r in ratios:
                a, b = full[n][r], local[n][r]
                assert (a != a and b != b) or a == b
        lin = torch.nn.Linear(64, 48, bias=True).half().cuda()
        from asvd4llm_amd.modules.svd_linear import SVDLinear
        father = torch.nn.Module()
        father.x = SVDLinear._from_factors(torch.randn(48, 8, device="cuda").half(), torch.randn(8, 64, device="cuda").half(), lin.bias.data, 8)
        assert parallel.exchange_factors([("x", father, "x", lin)], {"x": 0}, mode="all") == 0
        torch.cuda.synchronize()
        parallel.set_group(None)
        assert parallel.backend() == "gloo"
        dist.barrier()
    finally:
        parallel.set_group(None)
        dist.destroy_process_group()


@pytest.mark.gpu
def test_fisher_calibration_vs_reference_fixture(gpu, golden, tmp_path, monkeypatch):
    """calib_fisher_info (backward in torch, per-channel statistic in asvd_absstat_accum sq_mean) against the fisher_info the imported
    reference computed for the same weights and token ids (tests/golden/fisher.npz).  fp32 model; tolerance 1e-5 relative (the kernel
    sums the 32..80 rows of grad^2 in fp32 in a different order from torch's mean)."""
    from tests.test_oracle_golden import _fisher_fixture_model
    from asvd4llm_amd.act_aware_utils import calib_fisher_info
    monkeypatch.chdir(tmp_path)
    model, calib, want = _fisher_fixture_model(golden)
    model = model.cuda()
    calib_fisher_info(model, calib, use_cache=False)
    for n, m in model.named_modules():
        if isinstance(m, torch.nn.Linear):
            np.testing.assert_allclose(m.fisher_info.cpu().numpy(), want[n], rtol=1e-5, atol=1e-10, err_msg=n)
