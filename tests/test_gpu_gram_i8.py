"""The Gram matrix of the Cholesky-QR reduction on the int8 matrix pipe (csrc/gram_i8.h) against exact integer arithmetic on the CPU (numpy int64
as the CHECKER): the kernel must deliver the EXACT Gram matrix of the digitised matrix X~ (every column a 24-bit fixed-point number under its own
power-of-two scale), X~ must sit within 2^-24 of the column's largest entry of X, and the fp64-MFMA kernel (ASVD_GRAM_I8=0) must agree with the
plain fp64 product."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _digitise(X):
    """numpy restatement of colmaxexp_kernel + split_i8_kernel: X [m, n] fp32 -> (t int64 [m, n], E int [n])"""
    mx = np.abs(X).max(axis=0)
    f, e2 = np.frexp(mx.astype(np.float32))
    E = e2 + (f >= np.float32(127.0 / 128.0))
    t = np.rint(np.ldexp(X.astype(np.float64), (23 - E)[None, :])).astype(np.int64)
    t[:, mx == 0] = 0
    return t, E


def _panels(X, m_pad, nb):
    """[m, n] -> [nb][m_pad][32] fp32, zero padded"""
    m, n = X.shape
    P = torch.zeros(nb, m_pad, 32)
    for p in range(nb):
        w = min(32, n - 32 * p)
        if w > 0:
            P[p, :m, :w] = X[:, 32 * p:32 * p + w]
    return P


def _gram(gpu, Xs, mode, seg_rows=0, scratch_rows=None):
    from asvd4llm_amd import _lib as L
    lib = L.load(True)
    batch = len(Xs)
    m, n = Xs[0].shape
    m_pad, n_pad = (m + 31) // 32 * 32, (n + 63) // 64 * 64
    nb = n_pad // 32
    P = torch.stack([_panels(X, m_pad, nb) for X in Xs]).to(gpu)
    G = torch.full((batch, n_pad, n_pad), float("nan"), dtype=torch.float64, device=gpu)
    rows = scratch_rows or (m_pad + 63) // 64 * 64
    scratch = torch.empty(3 * n_pad * rows * batch, dtype=torch.int8, device=gpu)
    ex = torch.zeros(batch, n_pad, dtype=torch.int32, device=gpu)
    rc = lib.asvd_test_gram(ctypes.c_void_p(P.data_ptr()), m_pad * 32, nb * m_pad * 32, nb, m_pad, batch, mode, seg_rows, ctypes.c_void_p(G.data_ptr()),
                            ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.c_void_p(ex.data_ptr()),
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return G.cpu().numpy(), ex.cpu().numpy()


def _upper_blocks(n_pad):
    I = np.arange(n_pad)[:, None] // 32
    J = np.arange(n_pad)[None, :] // 32
    return I <= J


def _check_exact(gpu, Xs, **kw):
    G, ex = _gram(gpu, Xs, 1, **kw)
    m, n = Xs[0].shape
    for b, X in enumerate(Xs):
        Xn = X.numpy()
        t, E = _digitise(Xn)
        assert (ex[b, :n][np.abs(Xn).max(axis=0) > 0] == E[np.abs(Xn).max(axis=0) > 0]).all()
        # X~ within 2^-24 of 2^E (rint: half of the last digit) of X
        Xt = np.ldexp(t.astype(np.float64), (E - 23)[None, :])
        assert (np.abs(Xt - Xn.astype(np.float64)) <= np.ldexp(1.0, E - 24)[None, :]).all()
        exact = t.T @ t                                            # int64: |t| <= 2^23, m <= 2^17 rows
        assert np.abs(t).max() <= 127 * 65536
        want = np.ldexp(exact.astype(np.float64), (E[:, None] + E[None, :] - 46))
        got = G[b, :n, :n]
        up = _upper_blocks(G.shape[1])[:n, :n]
        d = np.sqrt(np.diag(want))
        scale = np.outer(d, d) + 1e-300
        err = (np.abs(got - want) / scale)[up].max()
        assert err <= 4e-16, err
        # padding columns: zero Gram entries
        upf = _upper_blocks(G.shape[1])
        assert (G[b][:, n:][upf[:, n:]] == 0).all()
    return G


def test_exact_gram_of_the_digitised_matrix(gpu):
    g = torch.Generator().manual_seed(0)
    Xs = [torch.randn(1024, 256, generator=g) * torch.logspace(0, -6, 256)[None, :] for _ in range(2)]
    Xs[1][:, 7] = 0.0                      # an all-zero column
    Xs[1][3, 9] = 37.5                      # one entry far above the rest of its column
    Xs[0][:, 11] = 1.9999999                # mantissa just under the 127/128 limit's other side (exponent bump)
    Xs[0][:, 12] = 2.0 ** -130              # denormal column
    _check_exact(gpu, Xs)


def test_asymmetric_blocks_and_ragged_sizes(gpu):
    """sizes that are no multiple of the 128-column block or the 64-row stage: 200 columns (7 panels incl. padding), 1000 rows"""
    g = torch.Generator().manual_seed(1)
    X = torch.randn(1000, 200, generator=g) * (1 + 50 * torch.rand(200, generator=g))[None, :]
    _check_exact(gpu, [X])


def test_row_segments_are_added_exactly(gpu):
    g = torch.Generator().manual_seed(2)
    X = torch.randn(1500, 192, generator=g)
    G1 = _check_exact(gpu, [X])
    G3 = _check_exact(gpu, [X], seg_rows=512)                 # 3 segments of 512 rows
    G2 = _check_exact(gpu, [X], scratch_rows=768)             # what a small scratch forces: 2 segments
    up = _upper_blocks(G1.shape[1])
    assert np.abs(G1 - G3)[0][up].max() <= 1e-12 and np.abs(G1 - G2)[0][up].max() <= 1e-12


def test_nan_and_inf_columns_poison_their_row_and_column_only(gpu):
    g = torch.Generator().manual_seed(3)
    X = torch.randn(256, 128, generator=g)
    X[5, 40] = float("nan")
    X[6, 70] = float("inf")
    G, _ = _gram(gpu, [X], 1)
    up = _upper_blocks(128)
    bad = np.zeros((128, 128), bool)
    bad[[40, 70], :] = True
    bad[:, [40, 70]] = True
    assert np.isnan(G[0][up & bad]).all() and np.isfinite(G[0][up & ~bad]).all()


def test_fp64_kernel_and_int8_path_agree_at_size(gpu):
    """4096 x 4096 with activation-like column scales: the two paths against each other, column-scaled (what the Cholesky sees)"""
    g = torch.Generator(device=gpu).manual_seed(4)
    X = torch.randn(4096, 4096, generator=g, device=gpu) * (0.02 * (1 + 30 * torch.rand(4096, generator=g, device=gpu) ** 8))[None, :]
    X = X.cpu()
    G64, _ = _gram(gpu, [X], 0)
    G8, _ = _gram(gpu, [X], 1)
    ref = (X.double().T @ X.double()).numpy()
    up = _upper_blocks(4096)
    d = np.sqrt(np.diag(ref))
    s = np.outer(d, d)
    assert (np.abs(G64[0] - ref) / s)[up].max() <= 1e-14
    # the int8 path is the exact Gram matrix of X~, |x~ - x| <= 2^-24 2^E per entry: scaled entries move by ~1e-8 (random signs over 4096 rows)
    assert (np.abs(G8[0] - ref) / s)[up].max() <= 2e-7


# ---- the long-side product Y = X Vr on the int8 matrix pipe (csrc/nn_gemm_i8.h) against the bf16 six-product kernel and fp64

def _long_side_err(gpu, m, n, graded, monkeypatch, i8):
    """columns of U sigma against the fp64 product (X s) V on the device, relative to sigma_1 and to the column's own sigma"""
    from asvd4llm_amd import ops
    monkeypatch.setenv("ASVD_NN_I8", "1" if i8 else "0")
    g = torch.Generator(device=gpu).manual_seed(9)
    W = torch.randn(m, n, generator=g, device=gpu) * 0.02
    s = (1 + 40 * torch.rand(n, generator=g, device=gpu) ** 8) if graded else None
    if graded:  # graded spectrum as well: sigma over three decades
        kk = min(m, n)
        U0 = torch.linalg.qr(torch.randn(m, kk, generator=g, device=gpu))[0]
        V0 = torch.linalg.qr(torch.randn(n, kk, generator=g, device=gpu))[0]
        W = (U0 * torch.logspace(0, -3, kk, device=gpu)) @ V0.T
    U, S, V, info = ops.svd(W, s)
    assert info.status == 0
    Ws = (W * s if s is not None else W).double()
    if m < n:  # wide: the long side is V = Ws^T U / sigma
        Ws, U, V = Ws.T, V, U
    Y = Ws @ V.double()
    US = U.double() * S.double()
    col = (US - Y).norm(dim=0)
    return (col / S[0].double()).max().item(), (col / S.double().clamp_min(1e-30)).max().item(), \
        (U.double().T @ U.double() - torch.eye(U.shape[1], dtype=torch.float64, device=gpu)).abs().max().item()


@pytest.mark.parametrize("shape,graded", [((1024, 1024), False), ((2048, 1024), True), ((1024, 2304), True), ((1000, 200), False)])
def test_long_side_product_int8_vs_bf16_vs_fp64(gpu, shape, graded, monkeypatch):
    m, n = shape
    a1, r1, o1 = _long_side_err(gpu, m, n, graded, monkeypatch, True)
    a0, r0, o0 = _long_side_err(gpu, m, n, graded, monkeypatch, False)
    # u_j sigma_j = X v_j: fp32-level relative to sigma_1 on both paths.  The int8 form keeps eight of the nine digit products and accumulates exactly;
    # what is left is the rounding of its fixed-point operands (2^-25 of the row / column maximum per entry).  Measured, worst column over sigma_1, int8 / bf16:
    # 1024^2 Gaussian 1.9e-7 / 4.6e-7, graded 2048 x 1024 4.5e-7 / 4.7e-7, graded 1024 x 2304 9.5e-8 / 5.1e-7, 1000 x 200 2.2e-7 / 2.1e-7.
    # (The first version kept six products: 7.3e-7 / 1.6e-6 / - / 6.6e-7, and 2.4e-5 on the near-diagonal matrix of the snapshot test below.)
    print(f"long-side product {shape} graded={graded}: int8 {a1:.2e} (own sigma {r1:.2e}, |U^T U - I| {o1:.2e}); bf16 {a0:.2e} ({r0:.2e}, {o0:.2e})")
    assert a1 <= 1e-6 and a0 <= 1e-6, (a1, a0)
    assert a1 <= 1.5 * a0 + 1e-7 and r1 <= 2.0 * r0 + 1e-7, (a1, a0, r1, r0)
    assert o1 <= max(1.5 * o0, 1e-4), (o1, o0)


def test_long_side_product_nan_row_and_column_scale_extremes(gpu, monkeypatch):
    """activation scales spanning 1e-12 .. 1e12 (the column exponents must cancel exactly), and a NaN entry poisons the output as it did"""
    from asvd4llm_amd import ops
    monkeypatch.setenv("ASVD_NN_I8", "1")
    g = torch.Generator(device=gpu).manual_seed(10)
    W = torch.randn(640, 256, generator=g, device=gpu)
    s = torch.logspace(-12, 12, 256, device=gpu)[torch.randperm(256, generator=g, device=gpu)]
    U, S, V, info = ops.svd(W, s)
    assert info.status == 0
    Ws = (W * s).double()
    res = ((Ws @ V.double() - U.double() * S.double()).norm(dim=0) / S[0].double()).max().item()
    assert res <= 2e-6, res


# ---- the coupling snapshot of the sparse sweeps on the int8 matrix pipe (csrc/snapshot_i8.h) against the fp16 three-product form

@pytest.mark.parametrize("shape,kind", [((1024, 1024), "flat"), ((2048, 1024), "graded"), ((1536, 1536), "graded"), ((1024, 1024), "near_diagonal")])
def test_snapshot_int8_and_fp16_forms_agree(gpu, shape, kind, monkeypatch):
    """same marks to within the pairs that sit at the threshold: same sweep count (+-1), same singular values, both converged.  near_diagonal: a
    spike per column over a floor 1e-3 below it — the values of the floor live in the low digits, where a truncated digit product would be noise"""
    from asvd4llm_amd import ops
    m, n = shape
    g = torch.Generator(device=gpu).manual_seed(21)
    W = torch.randn(m, n, generator=g, device=gpu) * 0.02
    s = 1 + 40 * torch.rand(n, generator=g, device=gpu) ** 8
    if kind == "graded":
        kk = min(m, n)
        U0 = torch.linalg.qr(torch.randn(m, kk, generator=g, device=gpu))[0]
        V0 = torch.linalg.qr(torch.randn(n, kk, generator=g, device=gpu))[0]
        W = (U0 * torch.logspace(0, -3, kk, device=gpu)) @ V0.T
    if kind == "near_diagonal":
        W = torch.diag(1 + torch.rand(n, generator=g, device=gpu)) + 1e-3 * torch.randn(m, n, generator=g, device=gpu)
        s = None
    out = {}
    for v in ("1", "0"):
        monkeypatch.setenv("ASVD_SNAP_I8", v)
        U, S, V, info = ops.svd(W, s)
        assert info.status == 0, info
        out[v] = (U, S, V, info)
    S1, S0 = out["1"][1].double(), out["0"][1].double()
    print(f"snapshot {shape} {kind}: sweeps int8 {out['1'][3].sweeps} fp16 {out['0'][3].sweeps}")
    assert abs(out["1"][3].sweeps - out["0"][3].sweeps) <= 1, (out["1"][3], out["0"][3])
    assert ((S1 - S0).abs().max() / S0[0]).item() <= 2e-6
    U1, V1 = out["1"][0].double(), out["1"][2].double()
    k = min(m, n) // 2
    eye = torch.eye(k, dtype=torch.float64, device=gpu)
    assert (V1[:, :k].T @ V1[:, :k] - eye).abs().max().item() <= 2e-5   # the rotated columns: orthogonal to the tolerance the snapshot enforces
    Ws = (W * s if s is not None else W).double()
    assert ((Ws @ V1[:, :k] - U1[:, :k] * S1[:k]).norm(dim=0) / S1[0]).max().item() <= 5e-6
