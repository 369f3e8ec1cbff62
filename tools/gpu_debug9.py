import sys, os
sys.path.insert(0, ".")
os.environ["ASVD_DEBUG"] = "1"
import torch
from asvd4llm_amd import ops
from bench import synth
dev = torch.device("cuda")
for seed in (233, 236, 237, 238):
    W, scal = synth(4096, 11008, seed)
    s = ops.make_scale(scal.to(dev), alpha=0.5)
    print("seed", seed, "s min/max", float(s.min()), float(s.max()), "scal zeros", int((scal == 0).sum()), flush=True)
    U, S, V, info = ops.svd(W.to(dev), s, k=512)
    print("  ->", info, flush=True)
