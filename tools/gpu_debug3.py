import sys, os, time, json
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from tests.test_gpu_svd import llm_like
dev = torch.device("cuda")
W, s = llm_like(4096, 4096)
Wd, sd = W.to(dev), s.to(dev)
ops.svd_profile(True)
U, S, V, info = ops.svd_batched([Wd], [sd], max_sweeps=int(os.environ.get("MAXSW", "3")))
prof = ops.svd_profile()
print(os.environ.get("ASVD_INNER"), info[0], {k: (round(v["ms"], 2), v["launches"], round(1e3 * v["ms"] / max(1, v["launches"]), 1)) for k, v in prof.items()})
