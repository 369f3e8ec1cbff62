import sys, os, time, json
sys.path.insert(0, ".")
import torch
from asvd4llm_amd import ops
from oracle import asvd_oracle as O
from tests.test_gpu_svd import llm_like, check_svd
dev = torch.device("cuda")
for shape in [(256, 256), (768, 3072), (1024, 1024)]:
    W, s = llm_like(*shape)
    r = O.rank_from_ratio(shape[0], shape[1], 0.9)
    print(shape, check_svd(dev, W, s, r), flush=True)
for batch in (1, 4):
    W, s = llm_like(4096, 4096)
    mats = [W.to(dev)] * batch; scs = [s.to(dev)] * batch
    ops.svd_batched(mats, scs)
    ops.svd_profile(True)
    torch.cuda.synchronize(); t0 = time.time()
    U, S, V, infos = ops.svd_batched(mats, scs)
    torch.cuda.synchronize(); dt = time.time() - t0
    prof = ops.svd_profile(); ops.svd_profile(False)
    print("batch", batch, "time", dt, infos[0], {k: (round(v["ms"], 2), v["launches"], round(1e3 * v["ms"] / max(1, v["launches"]), 1)) for k, v in prof.items()}, flush=True)
    if batch == 1:
        Ws = O.scaled_weight(W, s)
        Uo, So, Vo = O.exact_svd(Ws)
        r = 1843
        print("sigma top-r", O.sigma_rel_err(S[0].cpu(), So, r), "all", O.sigma_rel_err(S[0].cpu(), So, 4096))
        Ud, Vd, Sc = U[0].cpu().double(), V[0].cpu().double(), S[0].cpu().double()
        Rg = (Ud[:, :r] * Sc[:r]) @ Vd[:, :r].T
        Ro = (Uo[:, :r].double() * So[:r].double()) @ Vo[:, :r].double().T
        print("recon r", ((Rg - Ro).norm() / Ws.double().norm()).item(), "orthV", (Vd[:, :r].T @ Vd[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item(),
              "orthU", (Ud[:, :r].T @ Ud[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item(), flush=True)
        full = ((Ud * Sc) @ Vd.T - Ws.double()).norm() / Ws.double().norm()
        print("full recon", full.item(), "orthV full", (Vd.T @ Vd - torch.eye(4096, dtype=torch.float64)).abs().max().item())
