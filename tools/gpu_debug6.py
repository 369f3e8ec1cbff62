import sys, os, time, json
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from oracle import asvd_oracle as O
from tests.test_gpu_svd import llm_like
dev = torch.device("cuda")
for shape in ((4096, 4096), (11008, 4096), (4096, 11008)):
    m, n = shape
    W, s = llm_like(*shape)
    Ws = O.scaled_weight(W, s)
    Uo, So, Vo = O.exact_svd(Ws)
    r = O.rank_from_ratio(m, n, 0.9)
    Ro = (Uo[:, :r].double() * So[:r].double()) @ Vo[:, :r].double().T
    for tol in (1e-6, 1e-5, 3e-5):
        for k in (min(m, n), r):
            ops.svd(W.to(dev), s.to(dev), k=k, tol=tol)
            torch.cuda.synchronize(); t0 = time.time()
            U, S, V, info = ops.svd(W.to(dev), s.to(dev), k=k, tol=tol)
            torch.cuda.synchronize(); dt = time.time() - t0
            Ud, Vd, Sc = U.cpu().double(), V.cpu().double(), S.cpu().double()
            Rg = (Ud[:, :r] * Sc[:r]) @ Vd[:, :r].T
            print(f"{shape} tol={tol:g} k={k} t={dt:.3f} sweeps={info.sweeps} sigma_top_r={O.sigma_rel_err(S.cpu(), So, r):.2e} recon_r={((Rg - Ro).norm() / Ws.double().norm()).item():.2e} "
                  f"orthU={(Ud[:, :r].T @ Ud[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e} orthV={(Vd[:, :r].T @ Vd[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e}", flush=True)
