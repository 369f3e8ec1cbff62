import sys, os, time
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from oracle import asvd_oracle as O
from tests.test_gpu_svd import llm_like
dev = torch.device("cuda")
m = n = 4096
W, s = llm_like(m, n)
Ws = O.scaled_weight(W, s); Uo, So, Vo = O.exact_svd(Ws)
r = 1843
Ro = (Uo[:, :r].double() * So[:r].double()) @ Vo[:, :r].double().T
for env in ({}, {"ASVD_FORCE_REDUCE": "1"}, {"ASVD_FORCE_REDUCE": "1", "ASVD_R": "1"}):
    for kk in ("ASVD_FORCE_REDUCE", "ASVD_R"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    for B in (1, 8):
        ops.svd_batched([W.to(dev)] * B, [s.to(dev)] * B)
        ops.svd_profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        U, S, V, infos = ops.svd_batched([W.to(dev)] * B, [s.to(dev)] * B)
        torch.cuda.synchronize(); dt = time.time() - t0
        prof = ops.svd_profile(); ops.svd_profile(False)
        Ud, Vd, Sc = U[0].cpu().double(), V[0].cpu().double(), S[0].cpu().double()
        Rg = (Ud[:, :r] * Sc[:r]) @ Vd[:, :r].T
        print(f"{env} B={B} t/svd={dt/B:.3f} sweeps={infos[0].sweeps} sigma_top_r={O.sigma_rel_err(S[0].cpu(), So, r):.2e} recon_r={((Rg - Ro).norm() / Ws.double().norm()).item():.2e} "
              f"orthU={(Ud[:, :r].T @ Ud[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e} orthV={(Vd[:, :r].T @ Vd[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item():.2e} "
              f"prof={ {k: round(v['ms'], 1) for k, v in prof.items()} }", flush=True)
