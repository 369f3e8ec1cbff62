import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


def pytest_collection_modifyitems(config, items):
    # GPU tests must run on a GPU box only; on a CPU-only box they are deselected by `-m "not gpu"` — if someone runs them
    # anyway without a device they fail loudly (no silent skip, no CPU fallback).
    pass


@pytest.fixture(scope="session")
def golden():
    class G:
        def json(self, name):
            return json.load(open(os.path.join(GOLDEN, name)))

        def npz(self, name):
            return np.load(os.path.join(GOLDEN, name))

    return G()


@pytest.fixture(scope="session")
def built_lib():
    from asvd4llm_amd import build
    build.build(verbose=False)
    from asvd4llm_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch
    from asvd4llm_amd import _lib
    assert torch.cuda.is_available(), "GPU test selected but torch sees no device"
    _lib.load(require_device=True)
    os.environ["ASVD_STRICT"] = "1"
    # the CPU oracle (LAPACK gesdd through torch) is fastest at ~16 threads on the GPU boxes' 256-thread hosts: 2.4 s per 4096^2 SVD against 13.8 s
    # at 128 threads (bench.py's thread sweep, every round) — torch's default there is one thread per core
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return torch.device("cuda", 0)
