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
