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
