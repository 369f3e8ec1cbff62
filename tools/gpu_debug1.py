import sys, os, time, json
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from oracle import asvd_oracle as O
dev = torch.device("cuda")
# 1. sqrt / make_scale mismatch
g = torch.Generator().manual_seed(2)
scal = (32 * torch.randn(1000, generator=g).abs()).float(); scal[:3] = 0
ref = O.make_scale(scal, 0.5); got = ops.make_scale(scal.to(dev), alpha=0.5).cpu()
bad = (ref != got).nonzero().flatten()
print("make_scale fp32 mismatches:", bad.numel())
for i in bad[:5].tolist():
    print(i, scal[i].item(), ref[i].item(), got[i].item(), (scal[i]**0.5).item(), torch.sqrt(scal[i]).item(), (scal[i].double()**0.5).item())
sq = torch.sqrt(scal); pw = scal ** 0.5
print("cpu sqrt vs pow(0.5) mismatches:", (sq != pw).sum().item(), " gpu(+eps) vs cpu sqrt+eps:", ((sq + 1e-6) != got).sum().item())
# 2. convergence history on tall matrix
os.environ["ASVD_DEBUG"] = "1"
from tests.test_gpu_svd import llm_like
for (m, n) in ((11008, 4096), (4096, 4096)):
    W, s = llm_like(m, n)
    torch.cuda.synchronize(); t0 = time.time()
    _, S, _, info = ops.svd(W.to(dev), s.to(dev), want_vectors=False)
    torch.cuda.synchronize(); print(m, n, "values-only", info, time.time() - t0, flush=True)
    So = torch.linalg.svdvals(O.scaled_weight(W, s))
    r = O.rank_from_ratio(m, n, 0.9)
    print("sigma err top-r", O.sigma_rel_err(S.cpu(), So, r), "all", O.sigma_rel_err(S.cpu(), So, min(m, n)), flush=True)
ops.svd_profile(True)
W, s = llm_like(4096, 4096)
torch.cuda.synchronize(); t0 = time.time()
U, S, V, info = ops.svd(W.to(dev), s.to(dev))
torch.cuda.synchronize(); print("4096 full", info, time.time() - t0)
print(json.dumps(ops.svd_profile()))
So = torch.linalg.svdvals(O.scaled_weight(W, s))
print("sigma err top-r", O.sigma_rel_err(S.cpu(), So, 1843), "all", O.sigma_rel_err(S.cpu(), So, 4096))
