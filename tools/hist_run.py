import sys, os
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from bench import synth
dev = torch.device("cuda")
W, scal = synth(4096, 4096, 233); Wd = W.to(dev); s = ops.make_scale(scal.to(dev), alpha=0.5)
U, S, V, info = ops.svd(Wd, s)
print(info)
