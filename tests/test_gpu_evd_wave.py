"""The wave-local 64x64 symmetric eigen-solver (csrc/evd_wave.hip) alone, through its test hook: one wave per matrix, registers only.
Checked against numpy (eigenvalues, orthogonality, Q^T G Q), against the CPU prototype of the same data flow (tools/proto_evd_wave.py:
identical arithmetic up to the hardware rcp / rsq approximations), and for the bookkeeping the kernels' epilogues rely on (closed-form
diagonal, descending ranks with ties by index, column scales, off-diagonal measures)."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gram(rng, cond, n=64, rows=256):
    X = rng.standard_normal((rows, n)) * np.logspace(0, -np.log10(cond) / 2, n)[None, :]
    X = X + 0.3 * X @ np.linalg.qr(rng.standard_normal((n, n)))[0]
    return (X.T @ X).astype(np.float32)


def _run(gpu, G, sweeps):
    from asvd4llm_amd import _lib
    lib = _lib.load(True)
    B = G.shape[0]
    Gd = torch.from_numpy(G).to(gpu).contiguous()
    Q = torch.empty_like(Gd)
    Gout = torch.empty_like(Gd)
    diag = torch.empty((B, 64), dtype=torch.float32, device=gpu)
    cs = torch.empty((B, 64), dtype=torch.float32, device=gpu)
    rnk = torch.empty((B, 64), dtype=torch.int32, device=gpu)
    meas = torch.empty((B, 2), dtype=torch.float32, device=gpu)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.asvd_test_evd_wave(p(Gd), B, sweeps, p(Q), p(diag), p(rnk), p(cs), p(Gout), p(meas), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (Q, diag, rnk, cs, Gout, meas)]


def test_converged_solve_matches_numpy(gpu):
    rng = np.random.default_rng(0)
    G = np.stack([_gram(rng, c) for c in (1e1, 1e2, 1e3, 1e4, 1e5, 1e6) for _ in range(4)])
    Q, diag, rnk, cs, Gout, meas = _run(gpu, G, sweeps=8)
    for b in range(G.shape[0]):
        w = np.linalg.eigvalsh(G[b].astype(np.float64))[::-1]
        qd = Q[b].astype(np.float64)
        assert np.abs(np.sort(diag[b].astype(np.float64))[::-1] - w).max() <= 2e-5 * w[0]
        assert np.abs(qd.T @ qd - np.eye(64)).max() <= 5e-6            # product of unit-norm-corrected rotations
        assert np.abs(qd.T @ G[b].astype(np.float64) @ qd - np.diag(diag[b])).max() <= 3e-5 * w[0]
        # bookkeeping of the epilogues
        order = np.argsort(-diag[b].astype(np.float64), kind="stable")
        want = np.empty(64, np.int64)
        want[order] = np.arange(64)
        assert np.array_equal(rnk[b], want)
        assert np.allclose(cs[b], 1.0 / np.linalg.norm(qd, axis=0), rtol=1e-6)
        # the image the sweeps leave: its off-diagonal is what Q^T G Q says (the diagonal lives in `diag`)
        T = qd.T @ G[b].astype(np.float64) @ qd
        off = Gout[b].astype(np.float64) - np.diag(np.diag(Gout[b].astype(np.float64)))
        assert np.abs(off - (T - np.diag(np.diag(T)))).max() <= 3e-5 * w[0]


def test_measures_of_the_input(gpu):
    rng = np.random.default_rng(1)
    G = np.stack([_gram(rng, c) for c in (1e1, 1e3, 1e5)])
    Q, diag, rnk, cs, Gout, meas = _run(gpu, G, sweeps=0)
    for b in range(G.shape[0]):
        g = G[b].astype(np.float64)
        d = np.diag(g)
        A = np.abs(g) / np.sqrt(np.outer(d, d))
        np.fill_diagonal(A, 0)
        At = np.abs(g) / np.maximum.outer(d, d)
        np.fill_diagonal(At, 0)
        lead = np.zeros((64, 64), bool)
        lead[:32, :] = True   # the hook marks positions 0..31 as the leading panel
        lead[:, :32] = True
        assert abs(meas[b, 0] - A.max()) <= 1e-5 * A.max()
        assert abs(meas[b, 1] - At[lead].max()) <= 1e-5 * At[lead].max()
        assert np.array_equal(Q[b], np.eye(64, dtype=np.float32)) and np.array_equal(Gout[b], G[b])   # zero sweeps: nothing moves
    bad = G.copy()
    bad[1, 5, 7] = np.nan
    bad[2, 9, 3] = np.inf
    m = _run(gpu, bad, sweeps=0)[5]
    assert np.isfinite(m[0]).all() and np.isnan(m[1]).all() and np.isnan(m[2]).all()


def test_one_sweep_follows_the_cpu_prototype(gpu):
    """one inner sweep (what the SVD runs per visit of a pair) against the numpy emulation of the lane / register data flow"""
    spec = importlib.util.spec_from_file_location("proto_evd_wave", os.path.join(ROOT, "tools", "proto_evd_wave.py"))
    proto = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(proto)
    rng = np.random.default_rng(2)
    G = np.stack([_gram(rng, c) for c in (1e1, 1e3, 1e5)])
    Q, diag, rnk, cs, Gout, meas = _run(gpu, G, sweeps=1)
    for b in range(G.shape[0]):
        g, q, d, hist = proto.solve(G[b], sweeps=1)
        scale = float(np.abs(d).max())
        assert np.abs(diag[b] - d).max() <= 2e-4 * scale     # same rotations up to v_rcp / v_rsq (1 ulp) in the angles
        assert np.abs(Q[b] - q).max() <= 2e-3
        off_gpu = Gout[b] - np.diag(np.diag(Gout[b]))
        off_cpu = g - np.diag(np.diag(g))
        assert np.abs(off_gpu).max() <= 1.5 * np.abs(off_cpu).max() + 1e-6 * scale
        qd = Q[b].astype(np.float64)
        assert np.abs(qd.T @ qd - np.eye(64)).max() <= 3e-6


_COOP_SNIPPET = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, {root!r})
from asvd4llm_amd import ops
g = torch.Generator().manual_seed(11)
n, batch = {n}, {batch}
mats = [(torch.randn(n, n, generator=g) / n ** 0.5).cuda() for _ in range(batch)]
scales = [torch.rand(n, generator=g).add_(0.5).cuda() for _ in range(batch)]
U, S, V, infos = ops.svd_batched(mats, scales)
torch.cuda.synchronize()
np.savez({out!r}, S=torch.stack(S).cpu().numpy(), U=torch.stack(U).cpu().numpy(), V=torch.stack(V).cpu().numpy(),
         sweeps=np.array([i.sweeps for i in infos]), status=np.array([i.status for i in infos]))
"""


@pytest.mark.parametrize("n,batch", [(768, 3), (2048, 2)])
def test_cooperative_launch_is_bit_identical(gpu, tmp_path, n, batch):
    """launches with at most 256 super-pairs run four waves per solve (coop_sweep, evd_wave.hip): same arithmetic element for element as the
    wave-local sweep, so the whole SVD must come out bit-identical with ASVD_EVDQ=0 (wave-local forced) and ASVD_EVDQ=1 (cooperative forced)"""
    import subprocess
    import sys
    res = {}
    for mode in ("0", "1"):
        out = str(tmp_path / f"evdq{mode}.npz")
        env = dict(os.environ, ASVD_EVDQ=mode)
        subprocess.run([sys.executable, "-c", _COOP_SNIPPET.format(root=ROOT, n=n, batch=batch, out=out)], check=True, env=env, timeout=600)
        res[mode] = np.load(out)
    assert (res["0"]["status"] == 0).all() and np.array_equal(res["0"]["sweeps"], res["1"]["sweeps"])
    for k in ("S", "U", "V"):
        assert np.array_equal(res["0"][k].view(np.uint32), res["1"][k].view(np.uint32)), k


def _ring_emulation(G, visits=1):
    """fp64 emulation of the cross-only visit on the ring (evd_wave.hip, evdw_sweep_ring): positions interleaved (2k = column k of the first panel,
    2k + 1 = column k of the second), phase A pairs (2k, 2k + 1), phase B pairs (2k + 1, (2k + 2) % 64), the rotated pair swaps places
    (J = [[s, c], [c, -s]]), 16 phase pairs per visit.  Returns (Q with natural rows and position columns, diagonal per position, image in
    ring positions, the set of natural column pairs that met)."""
    nat = np.array([(p & 1) * 32 + (p >> 1) for p in range(64)])
    g = G.astype(np.float64)[np.ix_(nat, nat)]
    q = np.eye(64)
    who = list(nat)          # natural column sitting at every position
    met = set()
    for _ in range(16 * visits):
        for par in (0, 1):
            for k in range(32):
                p, r = (2 * k + par) % 64, (2 * k + par + 1) % 64
                a, d, b = g[p, p], g[r, r], g[r, p]
                met.add((min(who[p], who[r]), max(who[p], who[r])))
                if not (abs(b) > 1e-8 * np.sqrt(a * d)):
                    c, s = 1.0, 0.0
                else:
                    zeta = (d - a) / (2 * b)
                    t = np.sign(zeta) / (abs(zeta) + np.sqrt(zeta * zeta + 1)) if zeta != 0 else 1.0
                    c = 1 / np.sqrt(1 + t * t)
                    s = t * c
                M = np.array([[s, c], [c, -s]])
                g[[p, r], :] = M @ g[[p, r], :]
                g[:, [p, r]] = g[:, [p, r]] @ M
                q[:, [p, r]] = q[:, [p, r]] @ M
                who[p], who[r] = who[r], who[p]
    pos = np.array([2 * x if x < 32 else 2 * (x - 32) + 1 for x in range(64)])
    return q[pos, :], np.diag(g).copy(), g, met


def test_ring_visit_meets_every_cross_pair_once_and_follows_the_fp64_emulation(gpu):
    """the cross-only visit of the two-level sweeps (ASVD_RING): 32 phases on the ring of interleaved positions — every column of the first panel
    meets every column of the second exactly once, no two columns of one panel meet — against its fp64 emulation (same rotations up to the
    hardware rcp / rsq in the angles)"""
    rng = np.random.default_rng(5)
    G = np.stack([_gram(rng, c) for c in (1e1, 1e3, 1e5)])
    qe, de, ge, met = _ring_emulation(G[0])
    assert met == {(i, 32 + j) for i in range(32) for j in range(32)}     # 1024 cross pairs, nothing else
    Q, diag, rnk, cs, Gout, meas = _run(gpu, G, sweeps=-1)
    for b in range(G.shape[0]):
        qe, de, ge, _ = _ring_emulation(G[b])
        scale = float(np.abs(de).max())
        qd = Q[b].astype(np.float64)
        assert np.abs(qd.T @ qd - np.eye(64)).max() <= 3e-6
        assert np.abs(diag[b] - de).max() <= 2e-4 * scale
        assert np.abs(Q[b] - qe).max() <= 2e-3
        # the image it leaves (ring positions) is Q^T G Q of the rotations it made, off the diagonal
        nat = np.array([(p & 1) * 32 + (p >> 1) for p in range(64)])
        T = qd.T @ G[b].astype(np.float64) @ qd
        off = Gout[b].astype(np.float64) - np.diag(np.diag(Gout[b].astype(np.float64)))
        assert np.abs(off - (T - np.diag(np.diag(T)))).max() <= 3e-5 * scale
        assert np.abs(np.diag(T) - diag[b]).max() <= 3e-5 * scale
        # and the visit did its job: the cross couplings shrank (the couplings inside the panels are not its business)
        where = np.array([(nat[p] < 32) for p in range(64)])
        cross = np.outer(where, ~where)
        g0 = G[b].astype(np.float64)[np.ix_(nat, nat)]
        assert np.abs(off[cross]).max() < 0.5 * np.abs((g0 - np.diag(np.diag(g0)))[cross]).max()


@pytest.mark.parametrize("n,batch", [(768, 3), (2048, 2)])
def test_ring_visits_whole_svd_parity_and_forms_bit_identical(gpu, tmp_path, n, batch):
    """ASVD_RING=1: the two-level sweeps visit their sub-pairs cross-only.  The SVD must still meet the contract against LAPACK, and the wave-local and
    the cooperative form of the ring visit must agree bit for bit (ASVD_EVDQ=0 / 1)."""
    import subprocess
    import sys
    res = {}
    for mode in ("0", "1"):
        out = str(tmp_path / f"ring_evdq{mode}.npz")
        env = dict(os.environ, ASVD_EVDQ=mode, ASVD_RING="1")
        subprocess.run([sys.executable, "-c", _COOP_SNIPPET.format(root=ROOT, n=n, batch=batch, out=out)], check=True, env=env, timeout=600)
        res[mode] = np.load(out)
    assert (res["0"]["status"] == 0).all() and np.array_equal(res["0"]["sweeps"], res["1"]["sweeps"])
    for k in ("S", "U", "V"):
        assert np.array_equal(res["0"][k].view(np.uint32), res["1"][k].view(np.uint32)), k
    g = torch.Generator().manual_seed(11)
    mats = [(torch.randn(n, n, generator=g) / n ** 0.5) for _ in range(batch)]
    scales = [torch.rand(n, generator=g).add_(0.5) for _ in range(batch)]
    for b in range(batch):
        So = torch.linalg.svdvals(mats[b] * scales[b][None, :]).numpy()
        S = res["0"]["S"][b]
        assert np.abs(S - So).max() <= 1e-4 * So[0] and (np.abs(S - So) / So)[: n // 2].max() <= 1e-4
        U, V = res["0"]["U"][b].astype(np.float64), res["0"]["V"][b].astype(np.float64)
        k = n // 2
        assert np.abs(U[:, :k].T @ U[:, :k] - np.eye(k)).max() <= 2e-5 and np.abs(V[:, :k].T @ V[:, :k] - np.eye(k)).max() <= 2e-5
        W = (mats[b] * scales[b][None, :]).double().numpy()
        assert np.linalg.norm(W - (U * S[None, :]) @ V.T) <= 1e-4 * np.linalg.norm(W)
