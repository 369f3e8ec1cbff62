"""Kernel-level parity of the two-level sweeps' streaming kernels (twolevel.h) against plain fp64 products on the same panels (torch matmul as the
CHECKER): the update pass [X_S X_T] <- [X_S X_T] Qfin (split-bf16), and the fused update + next-step Gram kernel (split-fp16 with power-of-two
column scales)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(gpu, batch, D, R=1024, nb=32, near_identity=False, seed=0):
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
    rc = lib.asvd_test_supdate(ctypes.c_void_p(Xw.data_ptr()), R * 32, nb * R * 32, ns, D, R, 256, ctypes.c_void_p(Q.data_ptr()),
                               ctypes.c_void_p(flags.data_ptr()), ctypes.c_void_p(done.data_ptr()), ctypes.c_void_p(nupd.data_ptr()), R // 256, npairs,
                               batch, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert nupd.tolist() == [npairs - 1] * batch
    # per column: error relative to the column's own norm (graded columns must keep their relative accuracy)
    err = (Xw.double() - ref).norm(dim=2) / ref.norm(dim=2).clamp_min(1e-300)
    return err.max().item()


@pytest.mark.parametrize("D", [1, 2, 5, 7])
def test_supdate_vs_fp64(gpu, D):
    assert _run(gpu, batch=3, D=D) <= 2e-6


def test_supdate_graded_columns_near_identity(gpu):
    """late-sweep regime: column norms spanning 1e5, rotations of 1e-3 — every column keeps fp32-level relative accuracy"""
    assert _run(gpu, batch=2, D=3, near_identity=True) <= 2e-6


def _pair_of_slot(k, d):
    h = d.bit_length() - 1
    S = ((k >> h) << (h + 1)) | (k & ((1 << h) - 1))
    return S, S ^ d


def _run_supgram(gpu, batch, D, E, ns, R=512, m_pad=None, rows_per_wg=128, seed=0, rest=(1,), graded=0.0, near_identity=False, din_fudge=1.0):
    """one launch of the fused kernel: X <- X Qfin per super-pair of step D, and the six partial Gram tiles of every super-pair of
    step E from the UPDATED panels; both against fp64."""
    from asvd4llm_amd import _lib as L
    lib = L.load(True)
    nb = 2 * ns
    pw2 = 1 << (ns - 1).bit_length()
    npairs = pw2 // 2
    m_pad = R if m_pad is None else m_pad
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.randn(batch, nb, R, 32, generator=g) * 0.05
    if graded:  # column norms spanning 10^graded, panel by panel AND inside every panel
        X = X * torch.logspace(0, -graded, nb * 32).view(1, nb, 1, 32)
    if near_identity:
        Q = torch.linalg.qr(torch.eye(128) + 1e-3 * torch.randn(batch, npairs, 128, 128, generator=g))[0].contiguous()
    else:
        Q = torch.linalg.qr(torch.randn(batch, npairs, 128, 128, generator=g))[0].contiguous()
    X, Q = X.to(gpu), Q.to(gpu)
    flags = torch.ones(batch, npairs, 4, dtype=torch.int32, device=gpu)
    for k in rest:
        flags[:, k] = 0  # super-pairs at rest: must stay untouched, and still feed the next step's tiles
    done = torch.zeros(batch, dtype=torch.int32, device=gpu)
    nupd = torch.zeros(batch, dtype=torch.int32, device=gpu)
    ref = X.double().clone()
    n_upd = 0
    for k in range(npairs):
        S, T = _pair_of_slot(k, D)
        if T >= ns or k in rest:
            continue
        n_upd += 1
        idx = [2 * S, 2 * S + 1, 2 * T, 2 * T + 1]
        out = torch.cat([ref[:, i] for i in idx], dim=2) @ Q[:, k].double()
        for j, i in enumerate(idx):
            ref[:, i] = out[:, :, 32 * j:32 * j + 32]
    nchunks = R // rows_per_wg
    Gx = torch.full((batch, npairs, nchunks, 6, 1024), float("nan"), device=gpu)
    Xw = X.clone()
    # squared column norms of every super-pair of step D before the update, in the pair's column order (what the eigen-solve launch leaves behind
    # in the library; a little off on purpose: the carried norms are estimates) — garbage for pairs that do not exist
    nrm2 = (X.double() ** 2).sum(dim=2) * din_fudge  # [batch, nb, 32]
    Din = torch.full((batch, npairs, 128), float("nan"), device=gpu)
    for k in range(npairs):
        S, T = _pair_of_slot(k, D)
        if S >= ns:
            continue
        Din[:, k, :64] = nrm2[:, 2 * S:2 * S + 2].reshape(batch, 64).float()
        Din[:, k, 64:] = nrm2[:, 2 * T:2 * T + 2].reshape(batch, 64).float() if T < ns else 0.0
    vp = ctypes.c_void_p
    rc = lib.asvd_test_supgram(vp(Xw.data_ptr()), R * 32, nb * R * 32, ns, D, E, R, m_pad, rows_per_wg, vp(Q.data_ptr()), vp(flags.data_ptr()),
                               vp(Din.data_ptr()), vp(Gx.data_ptr()), vp(done.data_ptr()), vp(nupd.data_ptr()), nchunks, npairs, batch,
                               vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert nupd.tolist() == [n_upd] * batch
    err_x = ((Xw.double() - ref).norm(dim=2) / ref.norm(dim=2).clamp_min(1e-300)).max().item()
    err_g = 0.0
    order = [(0, 2), (0, 3), (1, 2), (1, 3), (0, 1), (2, 3)]
    for k in range(npairs):
        S, T = _pair_of_slot(k, E)
        if T >= ns:
            assert torch.isnan(Gx[:, k]).all()  # padding pair of the schedule: never written
            continue
        pn = [ref[:, 2 * S, :m_pad], ref[:, 2 * S + 1, :m_pad], ref[:, 2 * T, :m_pad], ref[:, 2 * T + 1, :m_pad]]
        got = Gx[:, k].double().sum(dim=1).view(batch, 6, 32, 32)
        for t, (a, b) in enumerate(order):
            want = pn[a].transpose(1, 2) @ pn[b]
            scale = (pn[a].norm(dim=1).unsqueeze(2) * pn[b].norm(dim=1).unsqueeze(1)).clamp_min(1e-300)  # |x_i| |y_j|: error as a cosine
            err_g = max(err_g, ((got[:, t] - want).abs() / scale).max().item())
    return err_x, err_g


@pytest.mark.parametrize("D,E", [(1, 2), (2, 3), (3, 4), (5, 6), (7, 8), (14, 15), (15, 1), (6, 3)])
def test_supgram_update_and_next_tiles_vs_fp64(gpu, D, E):
    err_x, err_g = _run_supgram(gpu, batch=2, D=D, E=E, ns=16)
    assert err_x <= 2e-6 and err_g <= 2e-6, (err_x, err_g)


@pytest.mark.parametrize("D,E", [(1, 2), (3, 4), (4, 5), (8, 9), (11, 12), (15, 1)])
def test_supgram_padded_schedule(gpu, D, E):
    """ns = 12 super-panels in a schedule padded to 16: quads with absent members, pairs that do not exist"""
    err_x, err_g = _run_supgram(gpu, batch=2, D=D, E=E, ns=12, rest=(0, 3))
    assert err_x <= 2e-6 and err_g <= 2e-6, (err_x, err_g)


def test_supgram_graded_columns_near_identity(gpu):
    """late-sweep regime: column norms spanning 1e5 (across and inside the panels), rotations of 1e-3 — every column keeps its relative accuracy
    (the split-fp16 arithmetic works on power-of-two scaled columns)"""
    err_x, err_g = _run_supgram(gpu, batch=2, D=3, E=4, ns=16, graded=5.0, near_identity=True)
    assert err_x <= 2e-6 and err_g <= 2e-6, (err_x, err_g)


def test_supgram_graded_columns_dense_rotation(gpu):
    err_x, err_g = _run_supgram(gpu, batch=2, D=5, E=6, ns=16, graded=4.0)
    assert err_x <= 2e-6 and err_g <= 2e-6, (err_x, err_g)


@pytest.mark.parametrize("fudge", [1.0 / 64.0, 64.0])
def test_supgram_tolerates_carried_norms_that_are_off(gpu, fudge):
    """the column scales come from CARRIED squared norms (estimates): off by 8x either way must not cost accuracy (3 bits of fp16 headroom
    above, 14 + 10 bits of range below)"""
    err_x, err_g = _run_supgram(gpu, batch=2, D=2, E=3, ns=16, din_fudge=fudge)
    assert err_x <= 2e-6 and err_g <= 2e-6, (err_x, err_g)


def test_supgram_gram_rows_stop_at_m_pad(gpu):
    """rows beyond m_pad (the accumulated right factor) are rotated but do not enter the Gram tiles"""
    err_x, err_g = _run_supgram(gpu, batch=1, D=2, E=3, ns=8, R=512, m_pad=256, rows_per_wg=128)
    assert err_x <= 2e-6 and err_g <= 2e-6, (err_x, err_g)


@pytest.mark.parametrize("ns,grouped", [(64, 1), (80, 1), (80, 0), (48, 1), (96, 1), (112, 1), (12, 1), (20, 1), (24, 1), (40, 1), (6, 1), (10, 1), (172, 1), (160, 1), (216, 1)])
def test_super_panel_schedule_meets_every_pair_once(gpu, ns, grouped):
    """the pair schedule of the two-level sweeps (XOR, padded XOR, grouped): the pairs of a super-step are disjoint and every pair of
    super-panels meets exactly once per sweep; the grouped order (groups of 2..16 super-panels, round-robin over the groups) reaches the minimum
    of ns - 1 super-steps whenever ns splits into an even number (<= 16) of power-of-two groups — 79 for the 13B shapes' 80 super-panels"""
    from asvd4llm_amd import _lib as L
    lib = L.load(True)
    out = torch.full((256 * 512,), -7, dtype=torch.int32, device=gpu)
    nsteps, npairs = ctypes.c_int(), ctypes.c_int()
    rc = lib.asvd_test_super_schedule(ns, grouped, ctypes.c_void_p(out.data_ptr()), out.numel(), ctypes.byref(nsteps), ctypes.byref(npairs))
    assert rc == 0
    tab = out[: nsteps.value * npairs.value].view(nsteps.value, npairs.value).cpu().tolist()
    seen = {}
    for row in tab:
        used = set()
        for code in row:
            if code < 0:
                continue
            S, T = code >> 16, code & 0xFFFF
            assert S < T < ns and S not in used and T not in used
            used |= {S, T}
            seen[(S, T)] = seen.get((S, T), 0) + 1
    assert len(seen) == ns * (ns - 1) // 2 and set(seen.values()) == {1}
    if grouped:
        expect = {80: 95, 48: 47, 96: 95, 112: 111, 12: 11, 20: 19, 24: 23, 40: 39, 160: 159, 64: 63, 6: 7, 10: 11, 172: 255, 216: 255}[ns]
        assert nsteps.value == expect, (ns, nsteps.value)
    if not grouped and ns == 80:
        assert nsteps.value == 127
