"""N>1 path on CPU: world_size-2 gloo processes run the sharding map + the sensitivity all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from asvd4llm_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        names = [f"layer{i}" for i in range(7)]
        shapes = [(320, 64), (64, 64), (176, 64), (64, 176), (64, 64), (64, 64), (176, 64)]
        ratios = [0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
        owner = parallel.lpt_assign([parallel.svd_flops(*s) for s in shapes], ws)
        local = {}
        for i, n in enumerate(names):
            if owner[i] == rank:
                local[n] = {r: 100.0 + i * 1.000000123 + r / 3.0 for r in ratios}  # values needing full double precision
        full = parallel.allgather_sensitivities(local, names, ratios, owner)
        q.put((rank, owner, full))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_allgather_sensitivities_world2():
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    res.sort()
    (_, own0, full0), (_, own1, full1) = res
    assert own0 == own1 and set(own0) == {0, 1}
    assert full0 == full1
    assert list(full0.keys()) == [f"layer{i}" for i in range(7)]
    for i in range(7):
        for r in [0.4, 0.5, 0.6, 0.7, 0.8, 0.9]:
            assert full0[f"layer{i}"][r] == 100.0 + i * 1.000000123 + r / 3.0  # bit exact through the fp64 wire format


def test_single_process_passthrough():
    names = ["a", "b"]
    full = parallel.allgather_sensitivities({"a": {0.5: 1.0}, "b": {0.5: 2.0}}, names, [0.5], [0, 0])
    assert full == {"a": {0.5: 1.0}, "b": {0.5: 2.0}}


class _TwoLinear(torch.nn.Module):
    """stand-in for SVDLinear on CPU (tests only): factors from the oracle"""

    def __init__(self, A, B):
        super().__init__()
        self.A, self.B = A, B

    def forward(self, x):
        return torch.nn.functional.linear(torch.nn.functional.linear(x, self.B), self.A)


def _oracle_from_linear(linear, param_ratio, act_aware=False, ic_split=1, oc_split=1, alpha=1, sigma_fuse="UV", rank_align=1):
    from oracle import asvd_oracle as O
    o = O.from_linear_oracle(linear.weight.data, getattr(linear, "scaling_diag_matrix", None), param_ratio, alpha=alpha, act_aware=act_aware,
                             sigma_fuse=sigma_fuse, rank_align=rank_align)
    return _TwoLinear(o["A"], o["B"])


def _sweep_worker(rank, ws, port, q, tmpdir):
    import contextlib, io, os as _os
    _os.chdir(tmpdir)
    if ws > 1:
        _os.environ["MASTER_ADDR"] = "127.0.0.1"
        _os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from asvd4llm_amd import sensitivity
        from asvd4llm_amd.modules.svd_linear import SVDLinear
        from tests.tiny_lm import TinyLM, default_args
        SVDLinear.from_linear = staticmethod(_oracle_from_linear)  # CPU stand-in: the sharding logic is what is under test
        model = TinyLM()
        g = torch.Generator().manual_seed(5)
        calib = [{"input_ids": torch.randint(0, 50, (1, 16), generator=g)} for _ in range(3)]
        for n, m in model.named_modules():
            if isinstance(m, torch.nn.Linear):
                m.scaling_diag_matrix = torch.rand(m.in_features, generator=g) + 0.1
        args = default_args(keep_svd_cache=False, prefactorize=False)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            sens = sensitivity.calib_sensitivity_ppl(model, calib, args, use_cache=False)
        evaluated = sum(1 for l in buf.getvalue().splitlines() if l.startswith("model.") or l.startswith("lm_head"))
        q.put((rank, sens, evaluated))
    finally:
        if ws > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_sweep_equals_single_process(tmp_path):
    """world_size 2 (gloo): each rank sweeps only the layers it owns; after the all-gather both hold the complete dict, in the
    reference's order, with exactly the values a single process computes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    (tmp_path / "w1").mkdir()
    p = ctx.Process(target=_sweep_worker, args=(0, 1, 0, q, str(tmp_path / "w1")))
    p.start()
    _, ref, ev1 = q.get(timeout=200)
    p.join(30)
    port = _free_port()
    procs = []
    for r in range(2):
        (tmp_path / f"w2_{r}").mkdir()
        procs.append(ctx.Process(target=_sweep_worker, args=(r, 2, port, q, str(tmp_path / f"w2_{r}"))))
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=200) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert ev1 == 15 * 6
    assert res[0][2] + res[1][2] == 15 * 6 and 0 < res[0][2] < 15 * 6  # the work was split, nothing evaluated twice
    for _, sens, _ in res:
        assert list(sens.keys()) == list(ref.keys())
        for name in ref:
            assert sens[name] == ref[name]  # bit-identical floats through the fp64 wire format


def _oracle_svdlinear(linear, param_ratio, act_aware=False, ic_split=1, oc_split=1, alpha=1, sigma_fuse="UV", rank_align=1):
    """CPU stand-in that returns a real SVDLinear (ALinear/BLinear modules) built from oracle factors"""
    from oracle import asvd_oracle as O
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    o = O.from_linear_oracle(linear.weight.data, getattr(linear, "scaling_diag_matrix", None), param_ratio, alpha=alpha, act_aware=act_aware,
                             sigma_fuse=sigma_fuse, rank_align=rank_align)
    bias = linear.bias.data if linear.bias is not None else None
    return SVDLinear._from_factors(o["A"].to(linear.weight.dtype), o["B"].to(linear.weight.dtype), bias, o["rank"])


def _dist_pipeline_worker(rank, ws, port, q, tmpdir, mode):
    import contextlib, io, os as _os
    _os.chdir(tmpdir)
    if ws > 1:
        _os.environ["MASTER_ADDR"] = "127.0.0.1"
        _os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from asvd4llm_amd import binary_search, sensitivity
        from asvd4llm_amd.modules.svd_linear import SVDLinear
        from tests.tiny_lm import TinyLM, default_args
        SVDLinear.from_linear = staticmethod(_oracle_svdlinear)
        SVDLinear.drop_factor_cache = staticmethod(lambda l: None)
        model = TinyLM()
        g = torch.Generator().manual_seed(5)
        calib = [{"input_ids": torch.randint(0, 50, (1, 16), generator=g)} for _ in range(3)]
        for n, m in model.named_modules():
            if isinstance(m, torch.nn.Linear):
                m.scaling_diag_matrix = torch.rand(m.in_features, generator=g) + 0.1
        args = default_args(keep_svd_cache=False, prefactorize=False, fused_sweep=False, param_ratio_target=0.7, gather_factors=mode,
                            offload_raw_to_cpu=False)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            sens = sensitivity.calib_sensitivity_ppl(model, calib, args, use_cache=False)
            binary_search.binary_search_truncation_rank(model, sens, calib, args)
        state = {k: v.detach().float().numpy().copy() for k, v in model.state_dict().items()}  # numpy: no shared-memory handles through the queue
        kinds = {n: type(m).__name__ for n, m in model.named_modules() if n in model._asvd_layers_min_ratio}
        q.put((rank, model._asvd_layers_min_ratio, kinds, state))
    finally:
        if ws > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["rank0", "all"])
def test_sharded_decomposition_completes_the_model(tmp_path, mode):
    """world_size 2 (gloo) through sweep + search + sharded decomposition + factor exchange: the saving rank (rank 0; every rank with
    mode 'all') holds EVERY selected layer as an SVDLinear, with exactly the factors a single process produces."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    (tmp_path / "w1").mkdir()
    p = ctx.Process(target=_dist_pipeline_worker, args=(0, 1, 0, q, str(tmp_path / "w1"), mode))
    p.start()
    _, ratios1, kinds1, state1 = q.get(timeout=300)
    p.join(30)
    assert sum(1 for k in kinds1.values() if k == "SVDLinear") >= 3
    port = _free_port()
    procs = []
    for r in range(2):
        (tmp_path / f"w2_{r}").mkdir()
        procs.append(ctx.Process(target=_dist_pipeline_worker, args=(r, 2, port, q, str(tmp_path / f"w2_{r}"), mode)))
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rk, ratios, kinds, state in res:
        assert ratios == ratios1
        if rk == 0 or mode == "all":
            assert kinds == kinds1, f"rank {rk} does not hold the complete compressed model"
            assert state.keys() == state1.keys()
            for k in state1:
                assert (state[k] == state1[k]).all(), k
        else:
            n_svd = sum(1 for k in kinds.values() if k == "SVDLinear")
            assert 0 < n_svd < sum(1 for k in kinds1.values() if k == "SVDLinear")  # a shard only


@pytest.mark.timeout(300)
def test_bench_gpus2_dry_run_spawns_two_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must launch two ranks itself and report n_gpus = 2 (CPU/gloo dry run:
    launch plumbing only — no kernels run without a GPU)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry_run", "--steps", "2"], env=env, capture_output=True,
                         text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["scaling"] == "weak"
    # the configs[3] leg: LPT shard of a model's Linears, ONE all-gather of the sensitivities on the process group, replicated search
    sm = rec["sharded_model"]
    assert sm["collective_world_size"] == 2 and sm["collective_backend"] == "gloo" and sm["plan_identical_on_all_ranks"] is True
    assert sum(sm["layers_per_rank"]) == sm["linears"] == 22 and min(sm["layers_per_rank"]) >= 1 and sm["load_flops_max_over_mean"] < 1.1
    assert sm["allgather_ms"] > 0 and 0.85 < sm["plan_param_ratio"] <= 1.0


def test_bench_model_linears_order_matches_reference_walk(golden):
    """bench.py's shape list of a Llama model is in the order the reference's sweep visits the module tree"""
    import bench
    names = [n for n, _, _ in bench.model_linears("llama-2-7b")]
    assert names == golden.json("search_extra.json")["llama7b_shaped"]["order"] and len(names) == 225


def _cache_worker(rank, ws, port, q, tmpdir, shard):
    """both ranks in ONE working directory: calibration data + hook pass with use_cache=True, twice (second pass = the load branch)"""
    import contextlib, io, os as _os
    _os.chdir(tmpdir)
    _os.environ["MASTER_ADDR"] = "127.0.0.1"
    _os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from asvd4llm_amd import act_aware_utils, datautils, ops
        from tests.tiny_lm import TinyLM
        # CPU stand-ins for the two hook kernels (the cache / collective plumbing is what is under test)
        ops.absstat_partial = lambda x2, method: x2
        def _fin(work, T, C, acc, method):
            if "abs_max" in method:
                torch.maximum(acc, work.abs().amax(0), out=acc)
            else:
                acc.add_(work.abs().mean(0))
        ops.absstat_finalize = _fin
        out = []
        for it in range(2):
            model = TinyLM()
            model.config._name_or_path = "tiny/lm"
            calib = datautils.get_calib_data("synthetic", None, "tiny/lm", 3, seqlen=16, seed=7, vocab_size=50)
            with contextlib.redirect_stderr(io.StringIO()):
                act_aware_utils.calib_input_distribution(model, calib, "abs_mean", use_cache=True, shard_samples=shard)
            stats = {n: m.scaling_diag_matrix.numpy().copy() for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)}
            out.append((torch.cat([c["input_ids"] for c in calib]).numpy().copy(), stats))  # numpy: tensors would travel as file descriptors
        files = sorted(_os.listdir("cache"))
        q.put((rank, out, files))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("shard", [False, True])
def test_cache_files_are_written_once_and_read_complete_world2(tmp_path, shard):
    """every rank used to torch.save the same cache files and load as soon as the name existed.  Now rank 0 writes (temporary
    name + rename), the others wait; `cache_exists` gives every rank rank 0's answer.  With --shard_calib a rank that ran fewer samples than its
    peer (3 samples over 2 ranks) still takes part in the all-reduce with a buffer of the same layout."""
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cache_worker, args=(r, ws, port, q, str(tmp_path), shard)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=200) for _ in range(ws)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, out0, files0), (_, out1, files1) = res
    assert files0 == files1 and len(files0) == 2 and not any(".tmp." in f for f in files0), files0
    for (ids_a, st_a), (ids_b, st_b) in zip(out0, out1):   # rank 0 vs rank 1, both passes
        assert (ids_a == ids_b).all()
        assert st_a.keys() == st_b.keys()
        for n in st_a:
            assert (st_a[n] == st_b[n]).all(), n
    for n in out0[0][1]:   # computed (first pass) vs loaded from the cache (second pass)
        assert (out0[0][1][n] == out0[1][1][n]).all()
        assert abs(out0[0][1][n]).sum() > 0


def _gather_worker(rank, ws, port, q):
    """exchange_factors(mode "rank0") with several owners: every owner holds SVDLinear-shaped factors (and one a plain-Linear fallback with a
    bias) of its layers; rank 0 must end with every layer, bit for bit, whatever the number of messages per peer"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import torch.nn as nn
        from asvd4llm_amd.modules.svd_linear import SVDLinear

        class Slot:
            pass

        shapes = [(48, 32), (32, 32), (64, 32), (32, 64), (32, 32), (40, 24), (24, 40), (32, 32), (16, 16), (48, 32), (32, 48)]
        owner = {f"l{i}": (i * 3 + 1) % ws for i in range(len(shapes))}   # uneven: some peers hold more layers than others
        items, want = [], {}
        for i, (o, n) in enumerate(shapes):
            g = torch.Generator().manual_seed(1000 + i)
            r = 4 + i
            A, B = torch.randn(o, r, generator=g), torch.randn(r, n, generator=g)
            bias = torch.randn(o, generator=g) if i % 4 == 0 else None
            raw = nn.Linear(n, o, bias=bias is not None)
            slot = Slot()
            if i == 5:   # the reference's fallback after a failed factorisation: a plain Linear travels as kind 0
                mod = nn.Linear(n, o, bias=False)
                mod.weight.data = torch.randn(o, n, generator=g)
                want[f"l{i}"] = ("lin", mod.weight.data.clone(), None, None)
            else:
                mod = SVDLinear._from_factors(A.clone(), B.clone(), None if bias is None else bias.clone(), r)
                want[f"l{i}"] = ("svd", A, B, bias)
            slot.m = mod if owner[f"l{i}"] == rank else raw
            items.append((f"l{i}", slot, "m", raw))
        got = parallel.exchange_factors(items, owner, mode="rank0")
        ok = True
        if rank == 0:
            for name, slot, _, _ in items:
                kind, a, b, bias = want[name]
                m = slot.m
                if kind == "svd":
                    ok = ok and isinstance(m, SVDLinear) and torch.equal(m.ALinear.weight.data, a) and torch.equal(m.BLinear.weight.data, b)
                    ok = ok and ((bias is None and m.ALinear.bias is None) or torch.equal(m.ALinear.bias.data, bias))
                else:
                    ok = ok and type(m) is nn.Linear and torch.equal(m.weight.data, a)
        q.put((rank, got, ok, sum(1 for v in owner.values() if v != 0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_factor_gather_from_several_owners_world4():
    ws = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(ws))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    rank0 = res[0]
    assert rank0[1] == rank0[3] and rank0[2], rank0        # rank 0 received every layer it does not own, bit for bit
    assert all(r[1] == 0 for r in res[1:])


def _save_cache_fail_worker(rank, ws, port, q, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        os.chdir(tmpdir)
        open("blocker", "w").write("a FILE where rank 0 wants a directory")     # makedirs("blocker") fails on rank 0
        outcome = "no error"
        try:
            parallel.save_cache({"x": 1}, "blocker/cache.pt")
        except Exception as e:  # noqa: BLE001
            outcome = type(e).__name__
        ok_path = os.path.join(tmpdir, "fine", "cache.pt")
        parallel.save_cache({"x": 2}, ok_path)                                # and the next write works, on every rank, without a stuck barrier
        q.put((rank, outcome, parallel.cache_exists(ok_path), parallel.load_cache(ok_path)["x"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_failed_cache_write_raises_on_every_rank(tmp_path):
    """rank 0 failing in makedirs / torch.save must not leave the other ranks in a barrier: the ok-flag is broadcast, every rank raises"""
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_save_cache_fail_worker, args=(r, ws, port, q, str(tmp_path))) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(ws))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][1] != "no error" and res[1][1] == "RuntimeError", res      # rank 0: its own OSError; rank 1: told by the flag
    assert all(r[2] and r[3] == 2 for r in res)
