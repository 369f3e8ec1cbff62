import sys, os, ctypes
sys.path.insert(0, ".")
os.environ["ASVD_DEBUG"] = "1"; os.environ["ASVD_DEBUG_TALL_STOP"] = "1"
import torch
from asvd4llm_amd import ops, _lib as L
from bench import synth
dev = torch.device("cuda")
lib = L.load(True)
m, n, B, k = 4096, 11008, 5, 512
W, scal = synth(m, n, 233)
Wd = W.to(dev); s = ops.make_scale(scal.to(dev), alpha=0.5)
nb = ctypes.c_size_t(); lib.asvd_svd_worksize(B, m, n, 1, ctypes.byref(nb))
work = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
S = [torch.empty(k, device=dev) for _ in range(B)]; U = [torch.empty(m, k, device=dev) for _ in range(B)]; V = [torch.empty(n, k, device=dev) for _ in range(B)]
arr = ctypes.c_void_p * B
info = (ctypes.c_int * (4 * B))()
rc = lib.asvd_svd_batched(B, arr(*[Wd.data_ptr()] * B), 0, m, n, n, arr(*[s.data_ptr()] * B), 1, arr(*[u.data_ptr() for u in U]), arr(*[x.data_ptr() for x in S]),
                          arr(*[v.data_ptr() for v in V]), k, 0, 0.0, ctypes.c_void_p(work.data_ptr()), work.numel(), info, None)
print("rc", rc, "work bytes", nb.value)
m_pad, n_pad, nbp = 11008, 4096, 128
bs = m_pad * 32 * nbp
Xp = work[: bs * 4 * B].view(torch.float32).view(B, nbp, m_pad, 32)
for b in range(B):
    print("Xp", b, float(Xp[b].abs().sum()), "equal to 0:", bool(torch.equal(Xp[b], Xp[0])))
off_g = ((bs * 4 * B + 255) // 256) * 256
G = work[off_g: off_g + n_pad * n_pad * 8 * B].view(torch.float64).view(B, n_pad, n_pad)
for b in range(B):
    dg = torch.diagonal(G[b])
    print("Gdiag", b, float(dg.min()), float(dg.max()), "nan", bool(torch.isnan(G[b].triu()).any()), "row0", G[b][0, :4].tolist())
