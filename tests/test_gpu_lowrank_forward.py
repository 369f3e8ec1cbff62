"""K10 fused SVDLinear forward (csrc/lowrank_forward.hip) against the reference's two-nn.Linear forward
(/root/reference/modules/svd_linear.py:105-109) and against fp64."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _two_linear(x, A, B, bias):
    """the reference forward: fp16 BLinear then fp16 ALinear (+bias)"""
    return nn.functional.linear(nn.functional.linear(x, B), A, bias)


def _exact(x, A, B, bias):
    """fp64 evaluation of the SAME rounding points: z rounded to fp16 after the first product"""
    z = (x.double() @ B.double().T).half().double()
    y = z @ A.double().T
    return y + bias.double() if bias is not None else y


@pytest.mark.parametrize("T,K,r,N,with_bias", [
    (1, 4096, 1843, 4096, False),     # decode, llama-7b attention projection at ratio 0.9 (rank not a multiple of anything)
    (16, 4096, 1024, 4096, True),
    (2, 4096, 1024, 4096, True),      # decode sizes take the fused GEMV pair
    (3, 11008, 1500, 4096, True),
    (4, 4096, 2686, 11008, False),
    (33, 4096, 2686, 11008, False),   # up_proj at 0.9: two token tiles, rank > 2048
    (7, 11008, 1500, 4096, True),     # down_proj: K = 172 * 64
    (256, 768, 345, 3072, True),      # opt-125m fc1, the largest token count the entry accepts
    (5, 64, 3, 70, True),             # smallest legal K, rank < one slice, N not a multiple of the tile
])
def test_fused_forward_vs_two_linears_and_fp64(gpu, T, K, r, N, with_bias):
    from asvd4llm_amd import ops
    g = torch.Generator(device="cuda").manual_seed(T * 7 + r)
    x = torch.randn(T, K, device="cuda", generator=g).half()
    B = (torch.randn(r, K, device="cuda", generator=g) / K ** 0.5).half()
    A = (torch.randn(N, r, device="cuda", generator=g) / r ** 0.5).half()
    bias = torch.randn(N, device="cuda", generator=g).half() if with_bias else None
    Ap, Bp, work = ops.lowrank_pack(A, B)
    y = ops.lowrank_forward(x, Ap, Bp, bias, work)
    y2 = ops.lowrank_forward(x, Ap, Bp, bias, work)  # the barrier words are reusable, and the result is deterministic
    assert torch.equal(y, y2)
    ref = _two_linear(x, A, B, bias)
    exact = _exact(x, A, B, bias)
    scale = exact.abs().max().item()
    err_fused = (y.double() - exact).abs().max().item() / scale
    err_ref = (ref.double() - exact).abs().max().item() / scale
    # both paths round z to fp16 and y to fp16 with fp32 accumulation: the fused kernel must be as accurate as hipBLASLt's two GEMMs
    # (z can differ by one fp16 ulp where the fp32 sums straddle a rounding boundary, which moves y by ~2^-11 |A| — hence 2x + 1 ulp).
    assert err_fused <= 2 * err_ref + 2 ** -10, (err_fused, err_ref)
    assert (y.float() - ref.float()).abs().max().item() <= 4e-3 * scale


def test_svdlinear_forward_dispatch(gpu, monkeypatch):
    """SVDLinear.forward takes the fused launch only when asked to (ASVD_FUSED_FORWARD=1 or module.fused_forward), for decode-sized
    fp16 inputs, when ALinear/BLinear are plain hook-free nn.Linear; repacks when a weight changes; one workspace per stream."""
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    torch.manual_seed(0)
    lin = nn.Linear(256, 192, bias=True).half().cuda()
    m = SVDLinear.from_linear(lin, 0.6, act_aware=False)
    x = torch.randn(2, 7, 256, device="cuda").half()
    with torch.no_grad():
        monkeypatch.delenv("ASVD_FUSED_FORWARD", raising=False)
        y_ref = m(x)  # default: the reference's two GEMMs (the fused path is opt-in)
        assert getattr(m, "_fused", None) is None
        monkeypatch.setenv("ASVD_FUSED_FORWARD", "1")
        y = m(x)  # decode-sized fp16 input -> one launch
        assert getattr(m, "_fused", None) is not None and y.shape == (2, 7, 192)
        assert (y.float() - y_ref.float()).abs().max().item() <= 4e-3 * y_ref.float().abs().max().item()
        m.ALinear.weight.mul_(2.0)  # in-place edit bumps _version: the padded copy must follow
        y2 = m(x)
        b = m.ALinear.bias.float()
        assert torch.allclose(y2.float() - b, 2 * (y.float() - b), rtol=0, atol=8e-3 * y_ref.float().abs().max().item())
        # a `.data` edit does not bump the version: documented, refresh_fused() is the remedy
        m.ALinear.weight.data.mul_(0.5)
        m.refresh_fused()
        y3 = m(x)
        assert torch.allclose(y3.float(), y.float(), rtol=0, atol=8e-3 * y_ref.float().abs().max().item())
        # a second stream gets its own barrier / intermediate workspace
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            y4 = m(x)
        st.synchronize()
        assert len(m._fused[3]) == 2 and torch.equal(y4, y3)
        # a forward hook on BLinear (e.g. a calibration hook of a re-calibrated compressed model) must keep firing: nn.Linear path
        seen = []
        h = m.BLinear.register_forward_hook(lambda mod, i, o: seen.append(1))
        m(x)
        assert seen == [1]
        h.remove()
        monkeypatch.delenv("ASVD_FUSED_FORWARD")
        m.refresh_fused()
        m.fused_forward = True  # per-module opt-in
        m(x)
        assert m._fused is not None
        big = torch.randn(300, 256, device="cuda").half()  # more tokens than the dispatch takes: nn.Linear path
        assert m(big).shape == (300, 192)
        m._fused = None
        m(torch.randn(17, 256, device="cuda").half())
        assert m._fused is None  # 17 tokens: two GEMMs, nothing packed
    xg = x.clone().requires_grad_(True)
    m(xg).sum().backward()  # autograd path stays on nn.Linear
    assert xg.grad is not None


def test_fused_forward_rejects_bad_arguments(gpu):
    from asvd4llm_amd import ops, _lib
    A = torch.zeros(64, 8, device="cuda").half()
    B = torch.zeros(8, 64, device="cuda").half()
    Ap, Bp, work = ops.lowrank_pack(A, B)
    with pytest.raises(AssertionError):
        ops.lowrank_forward(torch.zeros(257, 64, device="cuda").half(), Ap, Bp, None, work)
    lib = _lib.load(True)
    x = torch.zeros(4, 64, device="cuda").half()
    y = torch.empty(4, 64, device="cuda").half()
    rc = lib.asvd_lowrank_forward_f16(x.data_ptr(), 4, Bp.data_ptr(), Ap.data_ptr(), None, 64, 64, 64, y.data_ptr(), work.data_ptr(), 16, None)
    assert rc == _lib.ASVD_E_WORKSPACE if hasattr(_lib, "ASVD_E_WORKSPACE") else rc == -2
    with pytest.raises(ValueError):
        ops.lowrank_pack(torch.zeros(64, 8, device="cuda").half(), torch.zeros(8, 72, device="cuda").half())


def test_strict_check_counts_launches_per_workspace(gpu, monkeypatch):
    """the ASVD_STRICT give-up check fired on a process-global launch counter, so with many SVDLinear modules only the few that
    happened to make the global 64th call were ever checked.  The counter is per workspace now: EVERY module is checked at its own 64th launch,
    and check_fused_forward() reads the flag on demand."""
    import torch
    from asvd4llm_amd import ops
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    monkeypatch.setenv("ASVD_STRICT", "1")
    monkeypatch.delenv("ASVD_DEBUG", raising=False)
    g = torch.Generator().manual_seed(5)
    mods = []
    for _ in range(3):
        A = (torch.randn(256, 64, generator=g) * 0.05).half().to(gpu)
        B = (torch.randn(64, 128, generator=g) * 0.05).half().to(gpu)
        m = SVDLinear._from_factors(A, B, None, 64)
        m.fused_forward = True
        mods.append(m)
    checked = []
    real = ops.lowrank_check
    monkeypatch.setattr(ops, "lowrank_check", lambda w: (checked.append(w.data_ptr()), real(w))[1])
    x = torch.randn(2, 128, generator=g).half().to(gpu)
    with torch.no_grad():
        for it in range(64):
            for m in mods:       # interleaved: a global counter would have hit (it * 3 + k) & 63 == 0 for one module only
                m(x)
    works = [next(iter(m._fused[3].values())).data_ptr() for m in mods]
    assert sorted(checked) == sorted(works), (checked, works)     # each workspace exactly once, at ITS 64th launch
    checked.clear()
    for m in mods:
        m.check_fused_forward()
    assert sorted(checked) == sorted(works)
    assert all(ops._LOWRANK_CALLS[w] == 0 for w in works)
