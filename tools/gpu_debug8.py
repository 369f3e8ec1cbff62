import sys, os, time, json
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from oracle import asvd_oracle as O
from tests.test_gpu_svd import llm_like
dev = torch.device("cuda")
for shape in ((11008, 4096), (4096, 11008)):
    m, n = shape
    W, s = llm_like(*shape)
    if m < n:
        s = O.make_scale((32 * torch.randn(n, generator=torch.Generator().manual_seed(9)).abs()).half(), 0.5)
    Ws = O.scaled_weight(W, s)
    Uo, So, Vo = O.exact_svd(Ws)
    r = O.rank_from_ratio(m, n, 0.9)
    Ro = (Uo[:, :r].double() * So[:r].double()) @ Vo[:, :r].double().T
    for env in ({}, {"ASVD_RT": "1"}, {"ASVD_NO_REDUCE": "1"}):
        for kk in list(os.environ):
            if kk in ("ASVD_RT", "ASVD_NO_REDUCE"): del os.environ[kk]
        os.environ.update(env)
        ops.svd(W.to(dev), s.to(dev), k=r)
        ops.svd_profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        U, S, V, info = ops.svd(W.to(dev), s.to(dev), k=r)
        torch.cuda.synchronize(); dt = time.time() - t0
        prof = ops.svd_profile(); ops.svd_profile(False)
        Ud, Vd, Sc = U.cpu().double(), V.cpu().double(), S.cpu().double()
        Rg = (Ud[:, :r] * Sc[:r]) @ Vd[:, :r].T
        print(f"{shape} {env} t={dt:.3f} sweeps={info.sweeps} sigma_top_r={O.sigma_rel_err(S.cpu(), So, r):.2e} recon_r={((Rg - Ro).norm() / Ws.double().norm()).item():.2e} "
              f"orthU={(Ud[:, :r].T @ Ud[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e} orthV={(Vd[:, :r].T @ Vd[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e} "
              f"prof={ {k: round(v['ms'], 1) for k, v in prof.items()} }", flush=True)
