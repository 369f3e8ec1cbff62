import sys, os, time, json
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from oracle import asvd_oracle as O
from tests.test_gpu_svd import llm_like
dev = torch.device("cuda")
shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096x4096").split("x"))
W, s = llm_like(*shape)
Ws = O.scaled_weight(W, s)
Uo, So, Vo = O.exact_svd(Ws)
m, n = shape
for k in (512, O.rank_from_ratio(m, n, 0.9), min(m, n)):
    r = min(k, O.rank_from_ratio(m, n, 0.9))
    Ro = (Uo[:, :r].double() * So[:r].double()) @ Vo[:, :r].double().T
    torch.cuda.synchronize(); t0 = time.time()
    U, S, V, info = ops.svd(W.to(dev), s.to(dev), k=k)
    torch.cuda.synchronize(); dt = time.time() - t0
    Ud, Vd, Sc = U.cpu().double(), V.cpu().double(), S.cpu().double()
    Rg = (Ud[:, :r] * Sc[:r]) @ Vd[:, :r].T
    print(f"k={k} r={r} t={dt:.3f} {info} sigma_top_r={O.sigma_rel_err(S.cpu(), So, r):.2e} recon_r={((Rg - Ro).norm() / Ws.double().norm()).item():.2e} "
          f"orthU={(Ud[:, :r].T @ Ud[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e} orthV={(Vd[:, :r].T @ Vd[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e}", flush=True)
