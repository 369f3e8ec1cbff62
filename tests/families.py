"""Input families for the whole-SVD parity tests and for tools/bench_families.py (VERDICT r5 missing-3 / weak-2).

The reference runs on TRAINED checkpoints (asvd.py:23-27) whose weight spectra decay and whose activation statistics have heavy tails; the
`llm_like` family of tests/test_gpu_svd.py (Gaussian + outlier columns) has a flat Marchenko-Pastur spectrum.  Here the spectrum is
constructed, W = P diag(sigma) Q^T with Haar-random orthonormal P, Q (QR of a Gaussian, sign-fixed), and the statistics vector follows the
shape of either hook (act_aware_utils.py:64-74): `abs_mean` (half-normal x n_calib, 1 % of the channels x 30) or `abs_max` (heavier tail: 1 % of
the channels x 300).  alpha in {0.5, 1} (svd_linear.py:33 defaults to 1).  Everything is seeded torch code; nothing here is an oracle and nothing here imports one."""
import math

import torch

SPECTRA = ("pow05", "pow1", "geo3", "cluster8", "lowrank_noise")
STATS = ("abs_mean", "abs_max")


def spectrum(kind, k):
    i = torch.arange(1, k + 1, dtype=torch.float64)
    if kind == "pow05":
        return i ** -0.5
    if kind == "pow1":
        return i ** -1.0
    if kind == "geo3":   # geometric 1 -> 1e-3
        return 10.0 ** (-3.0 * (i - 1) / max(1, k - 1))
    if kind == "cluster8":   # eight distinct values, each k/8 times (exactly degenerate clusters before the activation scaling)
        vals = torch.tensor([1.0, 0.5, 0.25, 0.1, 0.05, 0.02, 0.01, 0.005], dtype=torch.float64)
        return vals[torch.clamp((i - 1) * 8 // k, max=7).long()]
    if kind == "lowrank_noise":   # rank k/4 (power law) — the 1e-4 noise floor is added to W itself in make()
        s = i ** -0.5
        s[k // 4:] = 0
        return s
    raise ValueError(kind)


def haar(rows, cols, gen, device="cpu"):
    """rows x cols with orthonormal columns (rows >= cols), Haar-distributed: Q of a Gaussian with the signs of diag(R) fixed"""
    if device == "cpu":
        G = torch.randn(rows, cols, generator=gen, dtype=torch.float64)
    else:  # input generation only (tools/bench_families.py): the QR runs wherever torch puts it
        G = torch.randn(rows, cols, generator=gen, dtype=torch.float64).to(device)
    Q, R = torch.linalg.qr(G)
    return Q * torch.sign(torch.diagonal(R)).unsqueeze(0)


def make_weight(kind, m, n, seed=0, device="cpu"):
    """fp32 W [m, n] = P diag(spectrum) Q^T, scaled so that its entries look like a trained layer's (|W|_F^2 = m n 0.02^2)"""
    g = torch.Generator().manual_seed(seed)
    k = min(m, n)
    sig = spectrum(kind, k).to(device)
    P = haar(m, k, g, device)
    Q = haar(n, k, g, device)
    W = (P * sig) @ Q.T
    if kind == "lowrank_noise":
        N = torch.randn(m, n, generator=g, dtype=torch.float64).to(device)
        W = W + 1e-4 * float(sig[0]) / math.sqrt(max(m, n)) * N   # noise singular values ~ 1e-4 sigma_1 (1 + sqrt(min/max))
    W = W * (0.02 * math.sqrt(m * n) / float(W.norm()))
    return W.float()


def make_stat(stat, n, seed=0, n_calib=32):
    """fp16 statistics vector [n] with the shape of the `abs_mean` / `abs_max` hook accumulators"""
    g = torch.Generator().manual_seed(seed + 7919)
    base = n_calib * torch.randn(n, generator=g).abs()
    k = max(1, int(0.01 * n))
    idx = torch.randperm(n, generator=g)[:k]
    if stat == "abs_mean":
        base[idx] *= 30
    elif stat == "abs_max":
        base = base + n_calib * 0.5          # a maximum over many tokens is never near zero
        base[idx] *= 300
    else:
        raise ValueError(stat)
    return base.clamp(max=60000.0).to(torch.float16)


def make(kind, stat, m, n, seed=0, device="cpu"):
    """(W fp32 [m, n], statistics fp16 [n] or None).  The caller turns the statistics into s = stat**alpha + 1e-6 (svd_linear.py:48-59) with the
    oracle (tests) or with the library's own asvd_make_scale (tools/bench_families.py)."""
    W = make_weight(kind, m, n, seed, device)
    return W, (None if stat is None else make_stat(stat, n, seed))
