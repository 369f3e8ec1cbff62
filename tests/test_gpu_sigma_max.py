"""GPU parity of the Lanczos sigma_max kernel (K4s, stable-rank sensitivity) against CPU torch.linalg.svdvals in fp64.
Bar: relative error <= 1e-4 (BASELINE sigma tolerance); measured errors are ~1e-6."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _ref(W):
    return torch.linalg.svdvals(W.double())[0].item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(768, 768), (3072, 768), (768, 3072), (100, 70), (70, 100), (33, 1), (1, 33), (257, 130)])
def test_sigma_max_shapes(gpu, dtype, shape):
    from asvd4llm_amd import ops
    g = torch.Generator().manual_seed(shape[0] * 7 + shape[1])
    W = (torch.randn(*shape, generator=g) * 0.02).to(dtype)
    sig, info = ops.sigma_max_batched([W.to(gpu)])
    assert info[0][0] == 0, info
    want = _ref(W.float())
    assert abs(sig[0].item() - want) <= TOL * want, (sig[0].item(), want, info)


def test_sigma_max_batched_llm_like_and_deterministic(gpu):
    from asvd4llm_amd import ops
    from tests.test_gpu_svd import llm_like
    mats = []
    for b in range(5):
        W, s = llm_like(1024, 1024, seed=10 + b)
        mats.append((W * s.float()[None, :]).half())
    dev = [w.to(gpu) for w in mats]
    sig, info = ops.sigma_max_batched(dev)
    sig2, _ = ops.sigma_max_batched(dev)
    for b, w in enumerate(mats):
        want = _ref(w.float())
        assert info[b][0] == 0
        assert abs(sig[b].item() - want) <= TOL * want
        assert sig[b].item() == sig2[b].item()  # fixed summation order: bit-reproducible
    single, _ = ops.sigma_max_batched([dev[3]])
    assert abs(single[0].item() - sig[3].item()) <= 1e-6 * sig[3].item()


def test_sigma_max_hard_cases(gpu):
    from asvd4llm_amd import ops
    g = torch.Generator().manual_seed(0)
    # two nearly equal leading singular values (gap 1e-5): any Ritz value in between is within tolerance
    Q1, _ = torch.linalg.qr(torch.randn(512, 512, generator=g, dtype=torch.float64))
    Q2, _ = torch.linalg.qr(torch.randn(512, 512, generator=g, dtype=torch.float64))
    sv = torch.linspace(1.0, 0.01, 512, dtype=torch.float64)
    sv[1] = sv[0] * (1 - 1e-5)
    W = ((Q1 * sv) @ Q2.T).float()
    sig, info = ops.sigma_max_batched([W.to(gpu)])
    assert info[0][0] == 0 and abs(sig[0].item() - 1.0) <= TOL
    # flat spectrum (all singular values equal) and rank one
    sig, info = ops.sigma_max_batched([(3.0 * Q1).float().contiguous().to(gpu)])
    assert info[0][0] == 0 and abs(sig[0].item() - 3.0) <= 3.0 * TOL
    u, v = torch.randn(300, 1, generator=g), torch.randn(1, 200, generator=g)
    R1 = u @ v
    sig, info = ops.sigma_max_batched([R1.to(gpu)])
    assert abs(sig[0].item() - _ref(R1)) <= TOL * _ref(R1)
    # zero matrix, NaN input, strided rows
    sig, info = ops.sigma_max_batched([torch.zeros(64, 48, device=gpu)])
    assert sig[0].item() == 0.0 and info[0][0] == 0
    bad = torch.randn(64, 64, generator=g)
    bad[3, 5] = float("nan")
    sig, info = ops.sigma_max_batched([bad.to(gpu)])
    assert info[0][0] == 2 and sig[0].item() != sig[0].item()
    big = torch.randn(200, 300, generator=g).half().to(gpu)
    view = big[:, :136]
    sig, info = ops.sigma_max_batched([view])
    want = _ref(view.float().cpu())
    assert abs(sig[0].item() - want) <= TOL * want


def test_sigma_max_llama_shapes_vs_jacobi(gpu):
    """4096x4096 / 11008x4096 / 4096x11008 fp16: Lanczos against the values-only k=1 Jacobi SVD of the same matrix."""
    from asvd4llm_amd import ops
    g = torch.Generator().manual_seed(5)
    for shape in [(4096, 4096), (11008, 4096), (4096, 11008)]:
        W = (torch.randn(*shape, generator=g) * 0.02).half().to(gpu)
        sig, info = ops.sigma_max_batched([W])
        _, S, _, _ = ops.svd(W, None, k=1, want_vectors=False)
        assert info[0][0] == 0
        assert abs(sig[0].item() - S[0].item()) <= 2e-5 * S[0].item(), (shape, sig[0].item(), S[0].item(), info)
