import sys, os
sys.path.insert(0, ".")
os.environ["ASVD_DEBUG"] = "1"
import torch
from asvd4llm_amd import ops
from bench import synth
dev = torch.device("cuda")
W, scal = synth(4096, 11008, 233)
Wd = W.to(dev); s = ops.make_scale(scal.to(dev), alpha=0.5)
for B in (4, 5, 8):
    print("batch", B, flush=True)
    U, S, V, infos = ops.svd_batched([Wd] * B, [s] * B, k=512)
    print("  ->", [i.sweeps for i in infos], flush=True)
W2, scal2 = synth(2048, 6000, 1)
W2d = W2.to(dev); s2 = ops.make_scale(scal2.to(dev), alpha=0.5)
for B in (4, 8):
    print("small batch", B, flush=True)
    U, S, V, infos = ops.svd_batched([W2d] * B, [s2] * B, k=512)
    print("  ->", [i.sweeps for i in infos], flush=True)
