"""Generate tests/golden/* by IMPORTING the reference (/root/reference, read-only) in the build container.

Run once here (`python oracle/make_golden.py`); the reference never travels to the GPU box, only these vectors do.
Fixtures are data (inputs + outputs of the reference's own functions), no reference source text.
`lm_eval` is absent in this image: the three modules that import it at top level get an empty stub — the hot path only
uses evaluate_utils.evaluate_perplexity, which has no lm_eval dependency."""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

for name in ("lm_eval", "lm_eval.base", "lm_eval.evaluator", "lm_eval.tasks"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["lm_eval.base"].BaseLM = object
sys.modules["lm_eval"].evaluator = sys.modules["lm_eval.evaluator"]
sys.modules["lm_eval"].tasks = sys.modules["lm_eval.tasks"]
sys.modules["lm_eval"].base = sys.modules["lm_eval.base"]
sys.path.insert(0, REF)

from modules.svd_linear import SVDLinear  # noqa: E402
import act_aware_utils  # noqa: E402
import sensitivity as ref_sensitivity  # noqa: E402
import binary_search as ref_binary_search  # noqa: E402
import evaluate_utils as ref_eval  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


# ---- F-rank ---------------------------------------------------------------------------------------------------
def _reference_rank(out_f, in_f, ratio, align):
    """rank the reference's SVDLinear.from_linear computes (svd_linear.py:39-44), obtained by CALLING it: the proxy Linear carries a weight that is a
    1-element tensor expanded to [out, in] (numel and shape right, 4 bytes of storage; `.float()` of an fp32 tensor is the tensor itself), the
    factorisation is replaced by a spy that records the rank it is asked for and fails, and the random nn.Linear the reference builds after a failed
    factorisation is created on the meta device (moving it to the CPU raises: caught here, the rank is already known)."""
    seen = {}

    def spy(w, q=6, niter=2, M=None):
        seen["q"] = q
        raise RuntimeError("rank probe")

    proxy = types.SimpleNamespace(weight=torch.zeros(1, 1).expand(out_f, in_f), in_features=in_f, out_features=out_f)
    stock = torch.svd_lowrank
    torch.svd_lowrank = spy
    try:
        with torch.device("meta"), contextlib.redirect_stdout(io.StringIO()):
            try:
                SVDLinear.from_linear(proxy, ratio, rank_align=align)
            except NotImplementedError:  # "Cannot copy out of meta tensor": the fallback Linear's .to(cpu)
                pass
    finally:
        torch.svd_lowrank = stock
    return int(seen["q"])


def gen_rank():
    shapes = [(768, 768), (3072, 768), (768, 3072), (50272, 768), (4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096),
              (5120, 5120), (13824, 5120), (5120, 13824), (32000, 5120), (64, 64), (176, 64), (64, 176)]
    ratios = [0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.25] + [0.1 * i for i in range(1, 20)]
    rows = []
    for (o, i) in shapes:
        for ratio in ratios:
            for align in (1, 128):
                rows.append([o, i, ratio, align, _reference_rank(o, i, ratio, align)])
    # and the real thing on small shapes: the rank of the module from_linear returns
    for (o, i) in [(64, 64), (176, 64), (64, 176)]:
        for ratio in ratios:
            lin = nn.Linear(i, o)
            with contextlib.redirect_stdout(io.StringIO()):
                m = SVDLinear.from_linear(lin, ratio)
            if isinstance(m, SVDLinear):
                exp = [r for r in rows if r[0] == o and r[1] == i and r[2] == ratio and r[3] == 1][0][4]
                assert m.truncation_rank == min(exp, min(o, i)) or m.truncation_rank == exp, (o, i, ratio, m.truncation_rank, exp)
    json.dump({"columns": ["out", "in", "ratio", "align", "rank"], "rows": rows}, open(os.path.join(OUT, "rank_table.json"), "w"))


# ---- F-hook ---------------------------------------------------------------------------------------------------
class OneLinear(nn.Module):
    def __init__(self, c, dtype):
        super().__init__()
        self.config = types.SimpleNamespace(_name_or_path="golden/one_linear")
        self.lin = nn.Linear(c, 8, bias=False).to(dtype)
        self.device = torch.device("cpu")

    def forward(self, x):
        return self.lin(x)


def gen_hook():
    out = {}
    cases = [("a", (1, 64, 64), torch.float16), ("b", (64, 176), torch.float32), ("c", (1, 512, 200), torch.float16),
             ("d", (1, 2048, 768), torch.float16)]
    g = torch.Generator().manual_seed(233)
    cwd = os.getcwd()
    for tag, shape, dt in cases:
        xs = [(torch.randn(shape, generator=g) * (1 + 3 * torch.rand(shape[-1], generator=g))).to(dt) for _ in range(4)]
        if tag == "c":
            xs[1][..., 5, 7] = float("nan")  # NaN handling of abs_max / abs_mean
        for method in ("abs_mean", "abs_max"):
            accs = []
            for nb in range(1, 5):
                model = OneLinear(shape[-1], dt)
                with tempfile.TemporaryDirectory() as td:
                    os.chdir(td)
                    os.makedirs("cache")
                    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                        act_aware_utils.calib_input_distribution(model, [{"x": x} for x in xs[:nb]], method, use_cache=False)
                    os.chdir(cwd)
                accs.append(npy(model.lin.scaling_diag_matrix))
            out[f"{tag}_{method}_acc"] = np.stack(accs)
        if tag != "d":
            out[f"{tag}_x"] = np.stack([npy(x) for x in xs])
        else:
            out["d_seed_note"] = np.array([233])
            out["d_x"] = np.stack([npy(x) for x in xs])[:, :, ::8, :]  # subsample rows for size; stats below are for d_x
    # for case d regenerate accumulators on the stored (subsampled) input so the fixture is self-contained
    xs = [torch.from_numpy(a) for a in out["d_x"]]
    for method in ("abs_mean", "abs_max"):
        accs = []
        for nb in range(1, 5):
            model = OneLinear(768, torch.float16)
            with tempfile.TemporaryDirectory() as td:
                os.chdir(td)
                os.makedirs("cache")
                with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                    act_aware_utils.calib_input_distribution(model, [{"x": x} for x in xs[:nb]], method, use_cache=False)
                os.chdir(cwd)
            accs.append(npy(model.lin.scaling_diag_matrix))
        out[f"d_{method}_acc"] = np.stack(accs)
    np.savez_compressed(os.path.join(OUT, "hook.npz"), **out)


# ---- F-svd ----------------------------------------------------------------------------------------------------
def exact_lowrank(w, q=6, niter=2, M=None):
    U, S, Vh = torch.linalg.svd(w, full_matrices=False)
    return U[:, :q], S[:q], Vh[:q].transpose(0, 1)


def gen_svd():
    out = {}
    meta = []
    g = torch.Generator().manual_seed(233)
    cases = []
    for (o, i) in [(64, 64), (176, 64), (64, 176), (256, 128)]:
        for dist in ("gauss", "outlier"):
            for dt in (torch.float16, torch.float32):
                cases.append((o, i, dist, dt))
    stock = torch.svd_lowrank
    for ci, (o, i, dist, dt) in enumerate(cases):
        W = torch.randn(o, i, generator=g) * 0.02
        scal = 4 * torch.randn(i, generator=g).abs()
        if dist == "outlier":
            W[:, torch.randperm(i, generator=g)[: max(1, i // 50)]] *= 20
            scal[torch.randperm(i, generator=g)[: max(1, i // 50)]] *= 30
            scal[0] = 0.0  # dead channel -> s = 1e-6 (fp16: 1.0133e-6)
        lin = nn.Linear(i, o, bias=(ci % 2 == 0)).to(dt)
        lin.weight.data = W.to(dt)
        lin.scaling_diag_matrix = scal.to(dt)
        for fuse in ("UV", "U", "V"):
            for ratio, alpha in ((0.5, 0.5), (0.9, 1.0)):
                torch.svd_lowrank = exact_lowrank
                with contextlib.redirect_stdout(io.StringIO()):
                    m = SVDLinear.from_linear(lin, ratio, act_aware=True, alpha=alpha, sigma_fuse=fuse)
                torch.svd_lowrank = stock
                assert isinstance(m, SVDLinear)
                key = f"c{ci}_{fuse}_{ratio}_{alpha}"
                out[key + "_A"] = npy(m.ALinear.weight.data)
                out[key + "_B"] = npy(m.BLinear.weight.data)
                # stock (randomized) reference: truncation error in the scaled norm, for the one-sided Eckart-Young check
                torch.manual_seed(233)
                with contextlib.redirect_stdout(io.StringIO()):
                    ms = SVDLinear.from_linear(lin, ratio, act_aware=True, alpha=alpha, sigma_fuse=fuse)
                s = (lin.scaling_diag_matrix ** alpha + 1e-6).float()
                Ws = lin.weight.data.float() * s.view(1, -1)
                rec = (ms.ALinear.weight.data.float() @ ms.BLinear.weight.data.float()) * s.view(1, -1)
                stock_err = float((Ws - rec).norm() / Ws.norm())
                meta.append({"key": key, "case": ci, "out": o, "in": i, "dist": dist, "dtype": str(dt).split(".")[-1], "fuse": fuse,
                             "ratio": ratio, "alpha": alpha, "rank": int(m.truncation_rank), "has_bias": lin.bias is not None,
                             "stock_scaled_trunc_err": stock_err})
        out[f"c{ci}_W"] = npy(lin.weight.data)
        out[f"c{ci}_scal"] = npy(lin.scaling_diag_matrix)
    np.savez_compressed(os.path.join(OUT, "svd_linear.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "svd_linear_meta.json"), "w"), indent=0)


# ---- F-svd-mid: opt-125m shapes (BASELINE configs[0]), inputs regenerated from a seed ---------------------------
def gen_svd_mid():
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from oracle import asvd_oracle as O
    out, meta = {}, []
    stock = torch.svd_lowrank
    captured = {}

    def capturing_lowrank(w, q=6, niter=2, M=None):
        U, S, Vh = torch.linalg.svd(w, full_matrices=False)
        captured["S"] = S.clone()
        return U[:, :q], S[:q], Vh[:q].transpose(0, 1)

    for ci, (o, i, ratio) in enumerate([(768, 768, 0.9), (3072, 768, 0.9), (768, 3072, 0.9), (768, 768, 0.4)]):
        seed = 7000 + ci
        W, scal = O.synth_linear_numpy(o, i, seed)
        lin = nn.Linear(i, o, bias=False).to(torch.float16)
        lin.weight.data = W
        lin.scaling_diag_matrix = scal
        torch.svd_lowrank = capturing_lowrank
        with contextlib.redirect_stdout(io.StringIO()):
            m = SVDLinear.from_linear(lin, ratio, act_aware=True, alpha=0.5, sigma_fuse="UV")
        torch.svd_lowrank = stock
        assert isinstance(m, SVDLinear)
        r = int(m.truncation_rank)
        X = torch.from_numpy(np.random.Generator(np.random.PCG64(99 + ci)).standard_normal((i, 16)).astype(np.float32))
        A, B = m.ALinear.weight.data.float(), m.BLinear.weight.data.float()
        out[f"m{ci}_probe_x"] = npy(X)
        out[f"m{ci}_probe_y"] = npy(A @ (B @ X))          # the reference's compressed layer applied to 16 probe vectors
        out[f"m{ci}_probe_wx"] = npy(W.float() @ X)
        out[f"m{ci}_sigma"] = npy(captured["S"])           # full fp32 spectrum of W*diag(s) as the reference's factorisation saw it
        meta.append({"case": ci, "out": o, "in": i, "ratio": ratio, "alpha": 0.5, "seed": seed, "rank": r,
                     "inputs_sha256": O.tensor_checksum(W, scal),
                     "ref_rel_err_unscaled": float((W.float() - A @ B).norm() / W.float().norm())})
    np.savez_compressed(os.path.join(OUT, "svd_mid.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "svd_mid_meta.json"), "w"), indent=0)


# ---- tiny model for order / search / stable-rank / ppl ---------------------------------------------------------
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


class TinyLM(nn.Module):
    """Llama-shaped module tree (names only matter); forward returns fixed pseudo-logits so evaluate_perplexity runs."""

    def __init__(self, d=32, f=80, n_layers=2, vocab=50, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.config = types.SimpleNamespace(_name_or_path="golden/tiny_lm")
        self.model = nn.Module()
        self.model.embed_tokens = nn.Embedding(vocab, d)
        self.model.layers = nn.ModuleList([Layer(d, f) for _ in range(n_layers)])
        self.lm_head = nn.Linear(d, vocab, bias=False)
        self.device = torch.device("cpu")

    def forward(self, input_ids=None, labels=None, **kw):
        h = self.model.embed_tokens(input_ids)
        for l in self.model.layers:
            a = l.self_attn
            h = h + a.o_proj(torch.tanh(a.q_proj(h)) * torch.sigmoid(a.k_proj(h)) + a.v_proj(h))
            m = l.mlp
            h = h + m.down_proj(torch.nn.functional.silu(m.gate_proj(h)) * m.up_proj(h))
        logits = self.lm_head(h)
        if labels is not None:  # the reference's calib_fisher_info calls model(input_ids=..., labels=...) and backpropagates out[0]
            return (nn.functional.cross_entropy(logits.view(-1, logits.size(-1)), labels.reshape(-1)), logits)
        return (logits,)


def tiny_state(model):
    return {k: npy(v) for k, v in model.state_dict().items()}


def gen_search():
    out = {}
    model = TinyLM()
    args = types.SimpleNamespace(scaling_method="abs_mean", alpha=0.5, n_calib_samples=3, calib_dataset="wikitext2", compress_kv_cache=False,
                                 rank_align=1, act_aware=True, sigma_fuse="UV", ppl_target=-1, param_ratio_target=0.8, kv_cache_ratio_target=-1)
    g = torch.Generator().manual_seed(5)
    calib = [{"input_ids": torch.randint(0, 50, (1, 16), generator=g)} for _ in range(3)]
    cwd = os.getcwd()
    stock = torch.svd_lowrank
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        os.makedirs("cache")
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            act_aware_utils.calib_input_distribution(model, calib, "abs_mean", use_cache=False)
        scal = {n: npy(m.scaling_diag_matrix) for n, m in model.named_modules() if isinstance(m, nn.Linear)}
        torch.svd_lowrank = exact_lowrank
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            sens = ref_sensitivity.calib_sensitivity_ppl(model, calib, args, use_cache=False)
        sweep_log = [l for l in buf.getvalue().splitlines() if l and not l.startswith("input_ids")]
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            sens_sr = ref_sensitivity.calib_sensitivity_stable_rank(model, calib, args, use_cache=False)
        # ppl of the raw tiny model
        ids = torch.cat([c["input_ids"] for c in calib], 0)
        ppl_raw = ref_eval.evaluate_perplexity(model, ids, 3)
        results = {}
        for tag, kw in (("ratio0.8", dict(param_ratio_target=0.8)), ("ratio0.6", dict(param_ratio_target=0.6)),
                        ("kv0.5", dict(compress_kv_cache=True, kv_cache_ratio_target=0.5))):
            m2 = TinyLM()
            for n, mod in m2.named_modules():
                if isinstance(mod, nn.Linear):
                    mod.scaling_diag_matrix = torch.from_numpy(scal[n])
            a2 = types.SimpleNamespace(**{**vars(args), **kw})
            if a2.compress_kv_cache:
                sd = {k: {0.1 * i: float(100 - 3 * i + (hash(k) % 7)) for i in range(1, 20)} for k in sens}
            else:
                sd = sens
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                ref_binary_search.binary_search_truncation_rank(m2, sd, calib, a2)
            trace = [l for l in buf.getvalue().splitlines() if l.startswith("low=") or l.startswith("===")]
            ranks = {n: (int(mod.truncation_rank) if isinstance(mod, SVDLinear) else -1) for n, mod in m2.named_modules()
                     if isinstance(mod, (SVDLinear,)) or (isinstance(mod, nn.Linear) and not n.endswith("ALinear") and not n.endswith("BLinear"))}
            ppl = ref_eval.evaluate_perplexity(m2, ids, 3)
            results[tag] = {"trace": trace, "ranks": ranks, "ppl_after": ppl, "sens": {k: {str(r): float(v) for r, v in d.items()} for k, d in sd.items()}}
        torch.svd_lowrank = stock
        os.chdir(cwd)
    order = list(sens.keys())
    json.dump({"order": order, "sweep_log": sweep_log, "sensitivity_ppl": {k: {str(r): float(v) for r, v in d.items()} for k, d in sens.items()},
               "sensitivity_stable_rank": {k: {str(r): float(v) for r, v in d.items()} for k, d in sens_sr.items()},
               "ppl_raw": ppl_raw, "search": results, "calib_ids": [c["input_ids"].tolist() for c in calib],
               "model": {"d": 32, "f": 80, "n_layers": 2, "vocab": 50, "seed": 0}},
              open(os.path.join(OUT, "tiny_lm.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(OUT, "tiny_lm_state.npz"), **tiny_state(TinyLM()), **{"scal::" + k: v for k, v in scal.items()})


def gen_search_extra():
    """tests/golden/search_extra.json (round 4): (1) the reference's ppl-TARGET search (binary_search.py:64-87) on the tiny LM, with the
    factorisation patched to the exact oracle for determinism, sensitivities and scaling vectors of the existing tiny fixture; (2) its ratio-target
    search on a Llama-2-7B-SHAPED module tree (225 Linears x 6 ratios = 1350 candidates with many ties): proxy Linears whose weights are
    1-element tensors expanded to the real shapes (numel right, no memory), from_linear replaced by a recorder — what is pinned is the search:
    stable sort with ties, bisection trace, the plan of the LAST PROBED cut."""
    out = {}
    tiny = json.load(open(os.path.join(OUT, "tiny_lm.json")))
    st = np.load(os.path.join(OUT, "tiny_lm_state.npz"))
    sens = {k: {float(r): v for r, v in d.items()} for k, d in tiny["sensitivity_ppl"].items()}
    calib = [{"input_ids": torch.tensor(ids)} for ids in tiny["calib_ids"]]
    ids = torch.cat([c["input_ids"] for c in calib], 0)
    stock = torch.svd_lowrank
    cwd = os.getcwd()
    res = {}
    for tag, target in (("ppl54.1", 54.1), ("ppl54.3", 54.3), ("ppl60", 60.0)):
        model = TinyLM()
        model.load_state_dict({k: torch.from_numpy(st[k]) for k in st.files if not k.startswith("scal::")})
        for n, mod in model.named_modules():
            if isinstance(mod, nn.Linear):
                mod.scaling_diag_matrix = torch.from_numpy(st["scal::" + n])
        args = types.SimpleNamespace(scaling_method="abs_mean", alpha=0.5, n_calib_samples=3, calib_dataset="wikitext2", compress_kv_cache=False,
                                     rank_align=1, act_aware=True, sigma_fuse="UV", ppl_target=target, param_ratio_target=-1, kv_cache_ratio_target=-1)
        torch.svd_lowrank = exact_lowrank
        buf = io.StringIO()
        try:
            with tempfile.TemporaryDirectory() as td:
                os.chdir(td)
                with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                    ref_binary_search.binary_search_truncation_rank(model, sens, calib, args)
        finally:
            torch.svd_lowrank = stock
            os.chdir(cwd)
        trace = [l for l in buf.getvalue().splitlines() if l.startswith("low=") or l.startswith("===")]
        ranks = {n: (int(mod.truncation_rank) if isinstance(mod, SVDLinear) else -1) for n, mod in model.named_modules()
                 if isinstance(mod, (SVDLinear,)) or (isinstance(mod, nn.Linear) and not n.endswith("ALinear") and not n.endswith("BLinear"))}
        res[tag] = {"ppl_target": target, "trace": trace, "ranks": ranks, "ppl_after": ref_eval.evaluate_perplexity(model, ids, 3)}
    out["tiny_ppl_target"] = res
    print("ppl-target traces:", {k: (len(v["trace"]), v["ppl_after"]) for k, v in res.items()})

    # ---- (2) Llama-2-7B-shaped search
    def proxy_linear(out_f, in_f):
        lin = nn.Linear(1, 1, bias=False)
        lin.in_features, lin.out_features = in_f, out_f
        lin.weight = nn.Parameter(torch.zeros(1, 1).expand(out_f, in_f), requires_grad=False)
        return lin

    class Blk(nn.Module):
        def __init__(self, h, f):
            super().__init__()
            self.self_attn = nn.Module()
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(self.self_attn, n, proxy_linear(h, h))
            self.mlp = nn.Module()
            self.mlp.gate_proj, self.mlp.up_proj, self.mlp.down_proj = proxy_linear(f, h), proxy_linear(f, h), proxy_linear(h, f)

    class Shaped(nn.Module):
        def __init__(self, h=4096, f=11008, layers=32, vocab=32000):
            super().__init__()
            self.model = nn.Module()
            self.model.layers = nn.ModuleList([Blk(h, f) for _ in range(layers)])
            self.lm_head = proxy_linear(vocab, h)

    def walk_order(model):  # the order the reference's sweep leaves in its dict: its own stack discipline (sensitivity.py:19-33) on this tree
        full = {m: n for n, m in model.named_modules()}
        order, modules = [], [model]
        while modules:
            sub = modules.pop()
            for name, child in sub.named_children():
                if isinstance(child, nn.Linear):
                    order.append(full[child])
                else:
                    modules.append(child)
        return order

    rng = np.random.RandomState(233)
    names = walk_order(Shaped())
    assert len(names) == 225
    sens7 = {}
    for li, n in enumerate(names):
        base = 5.5 + 0.9 * rng.rand()
        # two decimals: plenty of exact ties across layers and ratios (the stable sort keeps the traversal order among them)
        sens7[n] = {r: float(np.round(base + (0.9 - r) * (0.3 + 2.0 * rng.rand()) ** 2, 2)) for r in (0.4, 0.5, 0.6, 0.7, 0.8, 0.9)}
    vals = [v for d in sens7.values() for v in d.values()]
    assert len(vals) == 1350 and len(set(vals)) < 700, len(set(vals))
    runs = {}
    stock_from_linear = SVDLinear.from_linear
    for tag, target in (("ratio0.9", 0.9), ("ratio0.95", 0.95), ("ratio0.8", 0.8)):
        model = Shaped()
        picked = {}

        def recorder(linear, param_ratio, act_aware=False, ic_split=1, oc_split=1, alpha=1, sigma_fuse="UV", rank_align=1):
            m = nn.Identity()
            m.param_ratio = param_ratio
            return m

        SVDLinear.from_linear = staticmethod(recorder)
        args = types.SimpleNamespace(scaling_method="abs_mean", alpha=0.5, n_calib_samples=1, calib_dataset="wikitext2", compress_kv_cache=False,
                                     rank_align=1, act_aware=True, sigma_fuse="UV", ppl_target=-1, param_ratio_target=target, kv_cache_ratio_target=-1)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                ref_binary_search.binary_search_truncation_rank(model, sens7, [{"input_ids": torch.zeros(1, 4, dtype=torch.long)}], args)
        finally:
            SVDLinear.from_linear = stock_from_linear
        for n, mod in model.named_modules():
            if hasattr(mod, "param_ratio"):
                picked[n] = mod.param_ratio
        trace = [l for l in buf.getvalue().splitlines() if l.startswith("low=") or l.startswith("===")]
        runs[tag] = {"param_ratio_target": target, "trace": trace, "plan": picked}
    out["llama7b_shaped"] = {"shape": {"hidden": 4096, "inter": 11008, "layers": 32, "vocab": 32000}, "order": names,
                             "sens": {k: {str(r): v for r, v in d.items()} for k, d in sens7.items()}, "runs": runs}
    print("7B-shaped search:", {k: (len(v["trace"]), len(v["plan"])) for k, v in runs.items()})
    json.dump(out, open(os.path.join(OUT, "search_extra.json"), "w"), indent=0)


def gen_order_hf():
    """reverse-DFS Linear order of real HF module trees (tiny random Llama / OPT)"""
    from transformers import LlamaConfig, LlamaForCausalLM, OPTConfig, OPTForCausalLM
    res = {}
    for tag, model in (("llama", LlamaForCausalLM(LlamaConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                                              num_key_value_heads=2, vocab_size=64))),
                       ("opt", OPTForCausalLM(OPTConfig(hidden_size=32, ffn_dim=64, num_hidden_layers=2, num_attention_heads=2, vocab_size=64,
                                                        word_embed_proj_dim=32, max_position_embeddings=64)))):
        full = {m: n for n, m in model.named_modules()}
        order = []
        modules = [model]
        # the reference walk (sensitivity.py:19-33) executed through its own function would need a forward; the order only
        # depends on named_children(), so replay the identical stack discipline on the reference's data structure
        while modules:
            sub = modules.pop()
            for name, child in sub.named_children():
                if isinstance(child, nn.Linear):
                    order.append(full[child])
                else:
                    modules.append(child)
        res[tag] = order
    json.dump(res, open(os.path.join(OUT, "linear_order_hf.json"), "w"), indent=0)


# ---- F-fisher: calib_fisher_info (act_aware_utils.py:8-44) on the tiny LM ------------------------------------------
def gen_fisher():
    """fisher_info of every Linear after the reference's calib_fisher_info over 3 calibration batches (fp32 model), with the weights and
    token ids that produced it: pins oracle.fisher_update / fisher_finish and the build's calib_fisher_info (sq_mean hook kernel)."""
    model = TinyLM(seed=3)
    g = torch.Generator().manual_seed(17)
    calib = [{"input_ids": torch.randint(0, 50, (1, 24), generator=g)} for _ in range(3)]
    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stderr(io.StringIO()):
        cwd = os.getcwd()
        os.chdir(d)
        os.makedirs("cache")  # the reference saves cache/{model_id}_calib_fisher_info.pt unconditionally
        try:
            act_aware_utils.calib_fisher_info(model, calib, use_cache=False)
        finally:
            os.chdir(cwd)
    out = {"ids": np.stack([npy(c["input_ids"][0]) for c in calib])}
    for k, v in tiny_state(model).items():
        out["state::" + k] = v
    for name, m in model.named_modules():
        if isinstance(m, nn.Linear):
            out["fisher::" + name] = npy(m.fisher_info)
    np.savez_compressed(os.path.join(OUT, "fisher.npz"), **out)


def gen_hf_export():
    """F-export: what the reference's exported-repo loaders expect.  The reference's OWN model classes
    (huggingface_repos/modeling_asvd_{llama,opt}.py) are instantiated on tiny configs with a truncation_ranks dict; the state-dict keys /
    shapes they create, the config attribute they read, and the auto_map / architectures entries build_asvd_repo.py:66-88 writes are the
    fixture.  Under the installed transformers (5.x; the reference pins 4.41) ASVDLlamaConfig lacks three attributes newer LlamaModel
    code reads: the error is recorded, then the attributes are taken from a stock LlamaConfig of the same shape (data on the config
    INSTANCE; no reference file is touched) so that the class can still be instantiated."""
    import ast
    import importlib
    import transformers
    out = {"transformers_version": transformers.__version__, "families": {}}
    # auto_map / architectures literals, read out of the reference's exporter with ast (not retyped)
    src = open(os.path.join(REF, "huggingface_repos", "build_asvd_repo.py")).read()
    tree = ast.parse(src)
    assigns = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Constant) and node.test.left.value in ("opt", "llama"):
            fam = node.test.left.value
            for st in node.body:
                if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Subscript):
                    key = st.targets[0].slice.value if isinstance(st.targets[0].slice, ast.Constant) else None
                    if key in ("auto_map", "architectures"):
                        assigns.setdefault(fam, {})[key] = ast.literal_eval(st.value)
    assert set(assigns) == {"opt", "llama"} and all(set(v) == {"auto_map", "architectures"} for v in assigns.values()), assigns
    cases = {
        "llama": ("configuration_asvd_llama", "ASVDLlamaConfig", "modeling_asvd_llama", "ASVDLlamaForCausalLM", "LlamaConfig",
                  dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=64,
                       max_position_embeddings=64),
                  {"model.layers.0.self_attn.q_proj": 8, "model.layers.1.mlp.down_proj": 12, "lm_head": 10}),
        "opt": ("configuration_asvd_opt", "ASVDOPTConfig", "modeling_asvd_opt", "ASVDOPTForCausalLM", "OPTConfig",
                dict(hidden_size=32, ffn_dim=64, num_hidden_layers=2, num_attention_heads=2, vocab_size=64, word_embed_proj_dim=32,
                     max_position_embeddings=64),
                {"model.decoder.layers.0.self_attn.k_proj": 8, "model.decoder.layers.1.fc1": 12}),
    }
    for fam, (cfg_mod, cfg_cls, mdl_mod, mdl_cls, stock_cls, kw, ranks) in cases.items():
        cm = importlib.import_module("huggingface_repos." + cfg_mod)
        mm = importlib.import_module("huggingface_repos." + mdl_mod)
        rec = {"config_kwargs": kw, "truncation_ranks": ranks, "plain_instantiation_error": None, "config_attributes_added": []}
        cfg = getattr(cm, cfg_cls)(truncation_ranks=ranks, **kw)
        try:
            model = getattr(mm, mdl_cls)(cfg)
        except Exception as e:  # noqa: BLE001 — transformers-version drift of the pinned reference
            rec["plain_instantiation_error"] = f"{type(e).__name__}: {e}"
            stock = getattr(transformers, stock_cls)(**kw)
            for k, v in stock.to_dict().items():
                if not hasattr(cfg, k):
                    setattr(cfg, k, v)
                    rec["config_attributes_added"].append(k)
            model = getattr(mm, mdl_cls)(cfg)
        sd = model.state_dict()
        rec["state_dict"] = [[k, list(v.shape)] for k, v in sd.items()]
        rec["factor_module_class"] = sorted({type(m).__name__ for n, m in model.named_modules() if n in ranks})
        rec["config_reads"] = "truncation_ranks"
        rec["config_truncation_ranks_roundtrip"] = json.loads(json.dumps(cfg.to_dict()["truncation_ranks"]))
        rec["exporter_config_entries"] = assigns[fam]
        out["families"][fam] = rec
    json.dump(out, open(os.path.join(OUT, "hf_export_ref.json"), "w"), indent=1)
    print("hf_export_ref.json:", {f: (len(r["state_dict"]), r["plain_instantiation_error"]) for f, r in out["families"].items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "fisher":  # regenerate only the fisher fixture
        gen_fisher()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "hf_export":  # only the exported-repo fixture
        gen_hf_export()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "search_extra":  # ppl-target + model-scale search fixtures (needs tiny_lm.json / tiny_lm_state.npz)
        gen_search_extra()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rank":
        gen_rank()
        sys.exit(0)
    gen_rank()
    gen_hook()
    gen_svd()
    gen_svd_mid()
    gen_search()
    gen_search_extra()
    gen_order_hf()
    gen_fisher()
    gen_hf_export()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
