"""isolated test of the two-level update kernels (fp32 / split-bf16): X <- X Qfin on caller-built panels, several streams at once"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from asvd4llm_amd import _lib as L
lib = L.load(True)
fn = lib.asvd_dbg_supdate
_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
fn.restype = _i
fn.argtypes = [_i, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]
dev = torch.device("cuda")
torch.manual_seed(0)
R, nb, ns = 4096, 128, 64
npairs = 32
def run(split, nstreams, B=16, D=5, reps=3, scale_small=False):
    X = torch.randn(B, nb, R, 32, device=dev) * 0.05
    if scale_small:
        X *= torch.logspace(0, -5, nb, device=dev).view(1, nb, 1, 1)
    Q = torch.linalg.qr(torch.randn(B, npairs, 128, 128, device=dev))[0].contiguous()
    if scale_small:  # nearly the identity, like late sweeps
        Q = torch.linalg.qr(torch.eye(128, device=dev) + 1e-3 * torch.randn(B, npairs, 128, 128, device=dev))[0].contiguous()
    flags = torch.ones(B, npairs, 4, dtype=torch.int32, device=dev)
    done = torch.zeros(B, dtype=torch.int32, device=dev)
    # reference: pair k of step D: S = insert-zero-bit, T = S ^ D
    pairs = []
    h = D.bit_length() - 1
    for k in range(npairs):
        S = ((k >> h) << (h + 1)) | (k & ((1 << h) - 1)); pairs.append((S, S ^ D))
    ref = X.double().clone()
    for k, (S, T) in enumerate(pairs):
        idx = [2 * S, 2 * S + 1, 2 * T, 2 * T + 1]
        blk = torch.cat([ref[:, i] for i in idx], dim=2)  # [B, R, 128]
        out = blk @ Q[:, k].double()
        for j, i in enumerate(idx):
            ref[:, i] = out[:, :, 32 * j:32 * j + 32]
    worst = 0.0
    for rep in range(reps):
        Xw = X.clone()
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        torch.cuda.synchronize()
        per = [B // nstreams + (1 if g < B % nstreams else 0) for g in range(nstreams)]
        b0 = 0
        for g, st in enumerate(streams):
            nbg = per[g]
            rc = fn(split, Xw[b0].data_ptr(), R * 32, nb * R * 32, ns, D, R, 704, Q[b0].data_ptr(), flags[b0].data_ptr(), done[b0:].data_ptr(),
                    6, npairs, nbg, st.cuda_stream)
            assert rc == 0
            b0 += nbg
        torch.cuda.synchronize()
        err = ((Xw.double() - ref).abs().amax(dim=(1, 2, 3)) / ref.abs().amax(dim=(1, 2, 3)))
        worst = max(worst, err.max().item())
        bad = (err > 1e-5).nonzero().flatten().tolist()
        if bad:
            print("  rep", rep, "bad problems", bad, [f"{e:.2e}" for e in err.tolist()])
    print(f"split={split} streams={nstreams} small={scale_small}: worst rel err {worst:.3e}")
for small in (False, True):
    for split in (0, 1):
        for ns_ in (1, 3):
            run(split, ns_, scale_small=small)
