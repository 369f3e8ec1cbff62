import sys, os, time
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from tests.test_gpu_svd import llm_like
dev = torch.device("cuda")
W, s = llm_like(4096, 4096)
Wd, sd = W.to(dev), s.to(dev)
ops.svd(Wd, sd)
torch.cuda.synchronize(); t0 = time.time()
U, S, V, info = ops.svd(Wd, sd)
torch.cuda.synchronize(); print("inner", os.environ.get("ASVD_INNER"), "t", round(time.time() - t0, 3), info)
