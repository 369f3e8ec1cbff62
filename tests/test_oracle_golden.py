"""The oracle (oracle/asvd_oracle.py) replayed against vectors produced by the reference itself (oracle/make_golden.py).
CPU only.  This is what pins the oracle; the GPU parity tests then compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

from oracle import asvd_oracle as O


def test_rank_table(golden):
    t = golden.json("rank_table.json")
    for out_f, in_f, ratio, align, rank in t["rows"]:
        assert O.rank_from_ratio(out_f, in_f, ratio, align) == rank


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
@pytest.mark.parametrize("method", ["abs_mean", "abs_max"])
def test_hook_accumulator(golden, tag, method):
    g = golden.npz("hook.npz")
    xs = g[f"{tag}_x"]
    want = g[f"{tag}_{method}_acc"]
    acc = None
    acc_np = None
    for b in range(4):
        x = torch.from_numpy(xs[b])
        acc = O.hook_update(acc, x, method)
        np.testing.assert_array_equal(acc.numpy(), want[b])  # identical torch-CPU ops => bit exact (NaN positions included)
        # independent numpy restatement: equal up to one ulp of the activation dtype (fp32 vs fp64 column sums)
        acc_np = O.hook_update_numpy(acc_np, xs[b], method)
        a, w = acc_np.astype(np.float64), want[b].astype(np.float64)
        ok = np.isnan(w) == np.isnan(a)
        assert ok.all()
        fin = ~np.isnan(w)
        eps = np.finfo(want.dtype).eps
        assert np.all(np.abs(a[fin] - w[fin]) <= 2 * eps * np.abs(w[fin]) + 1e-30)


def test_from_linear_oracle_vs_reference(golden):
    meta = golden.json("svd_linear_meta.json")
    g = golden.npz("svd_linear.npz")
    for rec in meta:
        ci = rec["case"]
        dt = torch.float16 if rec["dtype"] == "float16" else torch.float32
        W = torch.from_numpy(g[f"c{ci}_W"])
        scal = torch.from_numpy(g[f"c{ci}_scal"])
        assert W.dtype == dt
        o = O.from_linear_oracle(W, scal, rec["ratio"], alpha=rec["alpha"], act_aware=True, sigma_fuse=rec["fuse"])
        assert o["rank"] == rec["rank"]
        A_ref = torch.from_numpy(g[rec["key"] + "_A"])
        B_ref = torch.from_numpy(g[rec["key"] + "_B"])
        assert o["A"].shape == A_ref.shape and o["B"].shape == B_ref.shape and o["A"].dtype == A_ref.dtype
        # sign / degenerate-subspace freedom of LAPACK across CPUs: compare the products, not the factors
        # (and only on numerically live channels in the unscaled norm: see oracle.live_channels)
        tol = 3e-3 if dt == torch.float16 else 2e-5
        e_live, e_scaled = O.recon_parity(o["A"], o["B"], A_ref, B_ref, W, o["s"])
        assert e_live < tol and e_scaled < tol
        # singular values carried by the factors (UV: |A_j| * |B_j s| ... use fuse U where A = U * S exactly)
        if rec["fuse"] == "U":
            sv = o["A"].double().norm(dim=0)
            sv_ref = A_ref.double().norm(dim=0)
            assert ((sv - sv_ref).abs() / sv_ref).max() < (2e-3 if dt == torch.float16 else 1e-5)
        # one-sided Eckart-Young check against the STOCK randomized reference call
        s = o["s"].float()
        Ws = W.float() * s.view(1, -1)
        rec_err = float((Ws - (o["A"].float() @ o["B"].float()) * s.view(1, -1)).norm() / Ws.norm())
        assert rec_err <= rec["stock_scaled_trunc_err"] * (1 + 1e-3) + (2e-3 if dt == torch.float16 else 1e-5)


def test_stable_rank_and_search_trace(golden):
    t = golden.json("tiny_lm.json")
    from tests.tiny_lm import load_golden_tiny
    model, scal = load_golden_tiny(golden)
    mods = dict(model.named_modules())
    # stable-rank sensitivities
    for name, d in t["sensitivity_stable_rank"].items():
        got = O.stable_rank_sensitivity(mods[name].weight.data)
        for r, v in d.items():
            assert abs(float(got[float(r)]) - v) <= 2e-6 * abs(v)
    # ratio-target binary search: trace lines and final per-layer ratios -> ranks
    for tag, kw in (("ratio0.8", dict(param_ratio_target=0.8)), ("ratio0.6", dict(param_ratio_target=0.6)),
                    ("kv0.5", dict(compress_kv_cache=True, kv_cache_ratio_target=0.5))):
        rec = t["search"][tag]
        sens = {k: {float(r): v for r, v in d.items()} for k, d in rec["sens"].items()}
        numel = {n: m.weight.numel() for n, m in mods.items() if isinstance(m, torch.nn.Linear)}
        ratios, trace = O.binary_search_ratios(sens, numel, **kw)
        ref_trace = [l for l in rec["trace"] if l.startswith("low=")]
        assert trace == ref_trace
        default = 2 if kw.get("compress_kv_cache") else 1
        for name, ratio in ratios.items():
            lin = mods[name]
            want_rank = rec["ranks"][name]
            if ratio == default:
                assert want_rank == -1
            else:
                assert O.rank_from_ratio(lin.out_features, lin.in_features, ratio) == want_rank


def test_mid_size_fixtures_pin_the_oracle(golden):
    """opt-125m shapes (BASELINE configs[0]): inputs regenerated from the seed, outputs produced by the imported reference."""
    meta = golden.json("svd_mid_meta.json")
    g = golden.npz("svd_mid.npz")
    for rec in meta:
        ci = rec["case"]
        W, scal = O.synth_linear_numpy(rec["out"], rec["in"], rec["seed"])
        assert O.tensor_checksum(W, scal) == rec["inputs_sha256"], "seed-regenerated fixture inputs differ from the ones the reference saw"
        o = O.from_linear_oracle(W, scal, rec["ratio"], alpha=rec["alpha"], act_aware=True, sigma_fuse="UV")
        r = rec["rank"]
        assert o["rank"] == r
        S_ref = torch.from_numpy(g[f"m{ci}_sigma"])
        assert O.sigma_rel_err(o["S"], S_ref, r) <= 1e-5
        X = torch.from_numpy(g[f"m{ci}_probe_x"])
        y = o["A"].float() @ (o["B"].float() @ X)
        y_ref, wx = torch.from_numpy(g[f"m{ci}_probe_y"]), torch.from_numpy(g[f"m{ci}_probe_wx"])
        assert torch.allclose(W.float() @ X, wx, rtol=1e-5, atol=1e-6)
        assert ((y - y_ref).norm() / wx.norm()).item() <= 1e-3


def _fisher_fixture_model(golden):
    from tests.tiny_lm import TinyLM
    fx = golden.npz("fisher.npz")
    model = TinyLM()
    model.load_state_dict({k[len("state::"):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("state::")})
    model.config._name_or_path = "tiny-fisher"
    calib = [{"input_ids": torch.from_numpy(row)[None]} for row in fx["ids"]]
    want = {k[len("fisher::"):]: fx[k] for k in fx.files if k.startswith("fisher::")}
    return model, calib, want


def test_fisher_restatement_vs_reference_fixture(golden):
    """oracle.fisher_update / fisher_finish replayed over the same three backward passes reproduce the fisher_info the imported
    calib_fisher_info (act_aware_utils.py:8-44) left on every Linear (tests/golden/fisher.npz, oracle/make_golden.py gen_fisher)."""
    model, calib, want = _fisher_fixture_model(golden)
    model.eval()
    acc = {n: 0 for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)}
    for batch in calib:
        ids = batch["input_ids"]
        model(input_ids=ids[:, :-1], labels=ids[:, 1:])[0].backward()
        for n, m in model.named_modules():
            if isinstance(m, torch.nn.Linear):
                acc[n] = O.fisher_update(acc[n], m.weight.grad)
        model.zero_grad()
    assert set(acc) == set(want) and len(want) == 15
    for n in want:
        got = O.fisher_finish(acc[n], len(calib)).numpy()
        np.testing.assert_allclose(got, want[n], rtol=2e-6, atol=1e-12, err_msg=n)
