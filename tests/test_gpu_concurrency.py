"""Concurrency contract of asvd_svd_batched (include/asvd_hip.h): calls from different host threads on different streams and
workspaces give exactly the results — bit for bit, sweep for sweep — of the same calls made one after the other, also while a third
stream keeps the GPU busy with foreign kernels (torch GEMMs) and while a call with a DIFFERENT pair schedule runs next to them.

Round 2 kept the pair schedules in __constant__ symbols that every call rewrote; two concurrent calls with different shapes could
overwrite each other's tables.  They are kernel arguments now (csrc/svd_jacobi.hip `Sched`); `test_mixed_schedules_concurrently`
is the regression test for that."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _problems(gpu, n_prob, m, n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_prob):
        W = torch.randn(m, n, generator=g) * 0.02
        W[:, torch.randperm(n, generator=g)[: max(1, n // 200)]] *= 20
        out.append(W.to(gpu))
    return out


def _same(a, b):
    """(U, S, V, infos) bit-identical, identical sweep counts"""
    for x, y in zip(a[:3], b[:3]):
        for t, u in zip(x, y):
            if not torch.equal(t, u):
                return False
    return [(i.status, i.sweeps) for i in a[3]] == [(i.status, i.sweeps) for i in b[3]]


def _run_concurrently(gpu, jobs, rounds, with_gemm_stream=True):
    """jobs: list of problem lists.  Every job runs `rounds` times in its own host thread on its own stream while (optionally) another
    thread streams torch GEMMs.  Returns per job the list of results of every round."""
    from asvd4llm_amd import ops
    results = [[] for _ in jobs]
    errors = []
    stop = threading.Event()
    barrier = threading.Barrier(len(jobs) + (1 if with_gemm_stream else 0))

    def svd_worker(i):
        try:
            st = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(st):
                barrier.wait()
                for _ in range(rounds):
                    results[i].append(ops.svd_batched(jobs[i]))
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    def gemm_worker():
        try:
            st = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(st):
                a = torch.randn(4096, 4096, device=gpu, dtype=torch.float16)
                b = torch.randn(4096, 4096, device=gpu, dtype=torch.float16)
                barrier.wait()
                while not stop.is_set():
                    for _ in range(8):
                        c = a @ b
                        a = (c * 1e-2).clamp_(-1, 1)
                    st.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=svd_worker, args=(i,)) for i in range(len(jobs))]
    tg = threading.Thread(target=gemm_worker) if with_gemm_stream else None
    if tg:
        tg.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    if tg:
        tg.join()
    torch.cuda.synchronize()
    assert not errors, errors
    return results


@pytest.mark.timeout(900)
def test_two_threads_two_streams_8x4096_plus_gemm_stream(gpu):
    """2 host threads x 2 streams x 8 x 4096^2 each + a third stream running torch GEMMs: bit-identical results AND identical sweep
    counts against the serial runs."""
    from asvd4llm_amd import ops
    jobs = [_problems(gpu, 8, 4096, 4096, seed=101), _problems(gpu, 8, 4096, 4096, seed=202)]
    ref = [ops.svd_batched(j) for j in jobs]
    torch.cuda.synchronize()
    assert all(i.status == 0 for r in ref for i in r[3])
    res = _run_concurrently(gpu, jobs, rounds=3)
    for i in range(2):
        assert len(res[i]) == 3
        for k, r in enumerate(res[i]):
            assert _same(r, ref[i]), f"thread {i} round {k}: differs from the serial run (sweeps {[x.sweeps for x in r[3]]} vs {[x.sweeps for x in ref[i][3]]})"


@pytest.mark.timeout(900)
def test_mixed_schedules_concurrently(gpu):
    """a grouped-schedule shape (5120 columns: 80 super-panels), an XOR shape (2048 columns), a padded-XOR shape (768 columns) and a
    single-level one (192 columns) in flight together: each call carries its own schedule in its kernel arguments"""
    from asvd4llm_amd import ops
    jobs = [_problems(gpu, 2, 5120, 5120, seed=1), _problems(gpu, 4, 2048, 2048, seed=2), _problems(gpu, 6, 768, 768, seed=3),
            _problems(gpu, 6, 300, 192, seed=4)]
    ref = [ops.svd_batched(j) for j in jobs]
    torch.cuda.synchronize()
    assert all(i.status == 0 for r in ref for i in r[3])
    res = _run_concurrently(gpu, jobs, rounds=4, with_gemm_stream=False)
    for i in range(len(jobs)):
        for k, r in enumerate(res[i]):
            assert _same(r, ref[i]), f"job {i} round {k} differs from its serial run"


@pytest.mark.timeout(900)
def test_split_batch_on_two_chip_halves_matches_the_unsplit_call(gpu, monkeypatch):
    """asvd_svd_batched runs a batch of >= 4 problems with >= 3072 columns as two halves, each on its own host thread and on a stream masked
    to one half of the CUs (csrc/svd_jacobi.hip, SplitCtx: the second half on the persistent worker thread of the calling thread).  Same singular values and vectors as the unsplit call up to fp32 rounding
    (the halves are planned for 128 CUs: other row splits, other summation orders), every problem converged, run-to-run deterministic,
    the info block of every problem filled, and ASVD_SPLIT=0 restores the one-stream call."""
    from asvd4llm_amd import ops
    n = 3072
    mats = _problems(gpu, 16, n, n, seed=77)
    monkeypatch.setenv("ASVD_SPLIT", "0")
    U0, S0, V0, i0 = ops.svd_batched(mats, k=1024)
    monkeypatch.setenv("ASVD_SPLIT", "1")
    U1, S1, V1, i1 = ops.svd_batched(mats, k=1024)
    U2, S2, V2, i2 = ops.svd_batched(mats, k=1024)
    assert all(i.status == 0 and 0 < i.sweeps <= 12 for i in i0 + i1)
    for b in range(16):
        assert torch.equal(S1[b], S2[b]) and torch.equal(U1[b], U2[b]) and torch.equal(V1[b], V2[b])     # deterministic
        rel = ((S1[b].double() - S0[b].double()).abs() / S0[b].double()).max().item()
        assert rel <= 2e-5, (b, rel)
        # same leading subspace (the 15 outlier columns of _problems stand clear of the bulk): all principal cosines are 1
        nout = max(1, n // 200)
        c = torch.linalg.svdvals((U0[b][:, :nout].double().T @ U1[b][:, :nout].double()).cpu())
        assert abs(1 - c.min().item()) <= 1e-5 and abs(1 - c.max().item()) <= 1e-5, (b, c.min().item(), c.max().item())   # fp32 vectors: orthonormal to ~1e-6
        idx = torch.arange(0, 64, device=gpu)
        eye = torch.eye(64, dtype=torch.float64, device=gpu)
        assert (V1[b][:, idx].double().T @ V1[b][:, idx].double() - eye).abs().max().item() <= 1e-5
    # a profiled call runs unsplit (per-class durations describe each kernel alone on the chip) and still gives the unsplit bits
    ops.svd_profile(True)
    U3, S3, V3, i3 = ops.svd_batched(mats, k=1024)
    prof = ops.svd_profile()
    ops.svd_profile(False)
    assert prof["supgram"]["launches"] > 0
    assert all(torch.equal(a, b) for a, b in zip(S3, S0))


@pytest.mark.timeout(600)
def test_split_batch_with_an_odd_number_of_problems(gpu, monkeypatch):
    """5 problems = halves of 3 and 2 (the info blocks, outputs and workspace slices of the second half start behind the first's)"""
    from asvd4llm_amd import ops
    n = 3072
    mats = _problems(gpu, 5, n, n, seed=91)
    monkeypatch.setenv("ASVD_SPLIT", "0")
    _, S0, _, i0 = ops.svd_batched(mats, k=256, want_vectors=False)
    monkeypatch.setenv("ASVD_SPLIT", "1")
    U1, S1, V1, i1 = ops.svd_batched(mats, k=256)
    assert all(i.status == 0 and 0 < i.sweeps <= 12 for i in i0 + i1)
    for b in range(5):
        assert ((S1[b].double() - S0[b].double()).abs() / S0[b].double()).max().item() <= 2e-5
        W = mats[b].double()
        r = (W @ V1[b].double() - U1[b].double() * S1[b].double()[None, :]).norm() / S1[b].double().norm()     # triplet residual |W V - U S|
        assert r.item() <= 2e-5, (b, r.item())


@pytest.mark.timeout(600)
def test_split_controls_path_bits_and_split_mode_profile(gpu, monkeypatch):
    """asvd_svd_set_split (the explicit switch next to the ASVD_SPLIT environment knob), the path bits a caller can read back
    (asvd_svd_get_last_path) and the split-mode profile (asvd_svd_set_profiling(2) + asvd_svd_get_split_profile): the halves report their own
    class times, the totals are their sums, the fused update + Gram launches of both halves lie on one time axis, and the bits of a call profiled
    that way are those of the plain split call."""
    from asvd4llm_amd import ops
    monkeypatch.delenv("ASVD_SPLIT", raising=False)
    n = 3072
    mats = _problems(gpu, 6, n, n, seed=55)
    U1, S1, V1, i1 = ops.svd_batched(mats, k=256)
    assert all(i.split and not i.split_refused and i.reduced and not i.reduce_fallback and not i.plain_retry for i in i1)
    try:
        ops.svd_set_split(0)
        _, S0, _, i0 = ops.svd_batched(mats, k=256)
        assert not any(i.split or i.split_refused for i in i0)
    finally:
        ops.svd_set_split(-1)
    _, _, _, ismall = ops.svd_batched(_problems(gpu, 4, 512, 512, seed=3))     # too small to qualify: neither split nor refused
    assert not any(i.split or i.split_refused for i in ismall) and all(i.reduced for i in ismall)
    ops.svd_profile(True, keep_split=True)
    try:
        U2, S2, V2, i2 = ops.svd_batched(mats, k=256)
        tot = ops.svd_profile()
        halves = ops.svd_split_profile()
    finally:
        ops.svd_profile(False)
    assert halves is not None and all(i.split for i in i2)
    assert all(torch.equal(a, b) for a, b in zip(S1, S2)) and all(torch.equal(a, b) for a, b in zip(U1, U2))   # profiling does not change the bits
    for c in ops.PROFILE_CLASSES:
        assert abs(tot[c]["ms"] - halves["halves"][0][c]["ms"] - halves["halves"][1][c]["ms"]) <= 1e-3 * max(1.0, tot[c]["ms"])
        assert tot[c]["launches"] == halves["halves"][0][c]["launches"] + halves["halves"][1][c]["launches"]
    assert halves["halves"][0]["supgram"]["launches"] > 0 and halves["halves"][1]["supgram"]["launches"] > 0
    ov = halves["supgram_ms"]
    assert abs(ov["half0"] - halves["halves"][0]["supgram"]["ms"]) <= 1e-2 * ov["half0"] and abs(ov["half1"] - halves["halves"][1]["supgram"]["ms"]) <= 1e-2 * ov["half1"]
    assert max(ov["half0"], ov["half1"]) <= ov["union"] * 1.001 and ov["union"] <= (ov["half0"] + ov["half1"]) * 1.001
    assert abs(ov["half0"] + ov["half1"] - ov["union"] - ov["both"]) <= 1e-2 * ov["union"]     # inclusion-exclusion on the common time axis
    # mode 1 (the default profile) still runs unsplit, and says so
    ops.svd_profile(True)
    try:
        _, _, _, i3 = ops.svd_batched(mats, k=256)
    finally:
        ops.svd_profile(False)
    assert not any(i.split for i in i3) and ops.svd_split_profile() is None
