"""Kernel-level parity of the two-level update pass (twolevel.h): [X_S X_T] <- [X_S X_T] Qfin for every super-pair of an XOR step, fp32
MFMA and split-bf16 arithmetic, against a plain fp64 product on the same panels (torch matmul as the CHECKER)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(gpu, split, batch, D, R=1024, nb=32, near_identity=False, seed=0):
    from asvd4llm_amd import _lib as L
    lib = L.load(True)
    ns, npairs = nb // 2, nb // 4
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = (torch.randn(batch, nb, R, 32, generator=g) * 0.05)
    if near_identity:  # graded columns and nearly-identity Q: the regime of the late sweeps
        X = X * torch.logspace(0, -5, nb).view(1, nb, 1, 1)
        Q = torch.linalg.qr(torch.eye(128) + 1e-3 * torch.randn(batch, npairs, 128, 128, generator=g))[0]
    else:
        Q = torch.linalg.qr(torch.randn(batch, npairs, 128, 128, generator=g))[0]
    X, Q = X.to(gpu), Q.contiguous().to(gpu)
    flags = torch.ones(batch, npairs, 4, dtype=torch.int32, device=gpu)
    flags[:, 1] = 0  # one super-pair of every problem is inactive: must stay untouched
    done = torch.zeros(batch, dtype=torch.int32, device=gpu)
    nupd = torch.zeros(batch, dtype=torch.int32, device=gpu)
    h = D.bit_length() - 1
    ref = X.double().clone()
    for k in range(npairs):
        S = ((k >> h) << (h + 1)) | (k & ((1 << h) - 1))
        T = S ^ D
        if k == 1:
            continue
        idx = [2 * S, 2 * S + 1, 2 * T, 2 * T + 1]
        blk = torch.cat([ref[:, i] for i in idx], dim=2)  # [B, R, 128]
        out = blk @ Q[:, k].double()
        for j, i in enumerate(idx):
            ref[:, i] = out[:, :, 32 * j:32 * j + 32]
    Xw = X.clone()
    rc = lib.asvd_test_supdate(split, ctypes.c_void_p(Xw.data_ptr()), R * 32, nb * R * 32, ns, D, R, 256, ctypes.c_void_p(Q.data_ptr()),
                               ctypes.c_void_p(flags.data_ptr()), ctypes.c_void_p(done.data_ptr()), ctypes.c_void_p(nupd.data_ptr()), R // 256, npairs,
                               batch, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert nupd.tolist() == [npairs - 1] * batch
    # per column: error relative to the column's own norm (graded columns must keep their relative accuracy)
    err = (Xw.double() - ref).norm(dim=2) / ref.norm(dim=2).clamp_min(1e-300)
    return err.max().item()


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("D", [1, 2, 5, 7])
def test_supdate_vs_fp64(gpu, split, D):
    assert _run(gpu, split, batch=3, D=D) <= 2e-6


@pytest.mark.parametrize("split", [0, 1])
def test_supdate_graded_columns_near_identity(gpu, split):
    """late-sweep regime: column norms spanning 1e5, rotations of 1e-3 — every column keeps fp32-level relative accuracy"""
    assert _run(gpu, split, batch=2, D=3, near_identity=True) <= 2e-6
