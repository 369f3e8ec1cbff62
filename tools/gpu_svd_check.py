"""GPU bring-up check for the SVD kernels (not a pytest; prints a table).  Compares against CPU torch.linalg.svd."""
import sys, time, json
import numpy as np
import torch
sys.path.insert(0, ".")
from asvd4llm_amd import ops

def llm_like(m, n, seed=233, n_calib=32, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(m, n, generator=g) * 0.02
    k = max(1, int(0.005 * n)); oc = torch.randperm(n, generator=g)[:k]; W[:, oc] *= 20
    scal = n_calib * torch.randn(n, generator=g).abs()
    k = max(1, int(0.01 * n)); oc = torch.randperm(n, generator=g)[:k]; scal[oc] *= 30
    scal = scal.to(torch.float16)
    s = scal ** 0.5 + 1e-6
    return W.to(dtype), s

def check(m, n, batch=1, time_it=True):
    W, s = llm_like(m, n)
    Wd, sd = W.cuda(), s.cuda()
    Ws = (W.float() * s.float().view(1, -1))
    t0 = time.time(); S64 = torch.linalg.svdvals(Ws.double()); tcpu64 = time.time() - t0
    kk = min(m, n)
    r = int(m * n * 0.9) // (m + n)
    ops.svd_profile(True)
    torch.cuda.synchronize(); t0 = time.time()
    U, S, V, infos = ops.svd_batched([Wd] * batch, [sd] * batch)
    torch.cuda.synchronize(); tg = time.time() - t0
    prof = ops.svd_profile()
    ops.svd_profile(False)
    if time_it:
        torch.cuda.synchronize(); t0 = time.time()
        ops.svd_batched([Wd] * batch, [sd] * batch)
        torch.cuda.synchronize(); tg2 = time.time() - t0
    else:
        tg2 = tg
    Sg = S[0].cpu().double()
    err_all = ((Sg - S64).abs() / S64).max().item()
    err_top = ((Sg[:r] - S64[:r]).abs() / S64[:r]).max().item()
    Ug, Vg = U[0].cpu().double(), V[0].cpu().double()
    orthU = (Ug[:, :r].T @ Ug[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item()
    orthV = (Vg[:, :r].T @ Vg[:, :r] - torch.eye(r, dtype=torch.float64)).abs().max().item()
    # rank-r reconstruction vs optimal
    Uo, So, Vho = torch.linalg.svd(Ws.double(), full_matrices=False)
    Ro = (Uo[:, :r] * So[:r]) @ Vho[:r]
    Rg = (Ug[:, :r] * Sg[:r]) @ Vg[:, :r].T
    rec = (Rg - Ro).norm().item() / Ws.double().norm().item()
    full = ((Ug * Sg) @ Vg.T - Ws.double()).norm().item() / Ws.double().norm().item()
    out = dict(m=m, n=n, batch=batch, info=str(infos[0]), sig_err_all=err_all, sig_err_top_r=err_top, r=r, orthU=orthU, orthV=orthV,
               recon_r_vs_oracle=rec, recon_full=full, t_first=tg, t_second=tg2, prof=prof)
    print(json.dumps(out), flush=True)
    return out

if __name__ == "__main__":
    shapes = [(64, 64), (128, 64), (64, 128), (100, 70), (256, 256), (512, 512), (768, 3072), (1024, 1024)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    res = []
    for (m, n) in shapes:
        res.append(check(m, n))
    json.dump(res, open("gpurun_out/svd_check.json", "w"), indent=1)
