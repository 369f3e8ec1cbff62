"""Sweeps and SVD/s per INPUT FAMILY at the bench size (default 32 x 4096^2): the flat `llm_like` family of bench.py next to constructed
spectra (tests/families.py: power laws, geometric, clustered, rank n/4 + noise) under alpha 0.5 / abs_mean- and alpha 1 / abs_max-shaped
statistics.  Sweep count IS the throughput of a Jacobi SVD, and every number of rounds 1-5 was measured on the flat family.

  python tools/bench_families.py [--batch 32] [--n 4096] [--m 4096] > profiles/r6_families.txt

One line per family: SVD/s of the default (split) call, sweeps (min..max over the batch), rotated pairs per sweep of a profiled (unsplit)
call, path bits (reduced / reduce_fallback / plain_retry), sigma of problem 0 and problem B-1 against torch.linalg.svdvals in fp64 ON THE DEVICE
(a checker, not the product path; the parity gate is tests/test_gpu_families.py against the CPU oracle)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--families", default="")
    ap.add_argument("--no_check", action="store_true")
    a = ap.parse_args()
    import torch
    from asvd4llm_amd import ops
    from tests import families as F
    dev = torch.device("cuda", 0)
    B, m, n = a.batch, a.m, a.n
    k = min(m, n)

    def llm_like(seed):
        g = torch.Generator().manual_seed(seed)
        W = torch.randn(m, n, generator=g) * 0.02
        kk = max(1, int(0.005 * n))
        W[:, torch.randperm(n, generator=g)[:kk]] *= 20
        scal = 32 * torch.randn(n, generator=g).abs()
        kk = max(1, int(0.01 * n))
        scal[torch.randperm(n, generator=g)[:kk]] *= 30
        return W, scal.half()

    cases = [("llm_like", "abs_mean", 0.5)]
    for kind in F.SPECTRA:
        cases += [(kind, "abs_mean", 0.5), (kind, "abs_max", 1.0)]
    if a.families:
        keep = set(a.families.split(","))
        cases = [c for c in cases if c[0] in keep]
    print(f"# {B} x {m}x{n} fp32 per call; SVD/s of the default call; sweeps / rotated pairs from a profiled (unsplit) call", flush=True)
    for kind, stat, alpha in cases:
        mats, stats = [], []
        t0 = time.time()
        if kind == "llm_like":
            for b in range(B):
                W, st = llm_like(1000 + b)
                mats.append(W.to(dev))
                stats.append(st.to(dev))
        else:
            # B distinct matrices from two Haar factors: W_b = (P[perm_b] diag(sigma)) Q[perm'_b]^T — row permutations of a Haar matrix are Haar
            g = torch.Generator().manual_seed(77)
            P = F.haar(m, k, g, dev)
            Q = F.haar(n, k, g, dev)
            sig = F.spectrum(kind, k).to(dev)
            for b in range(B):
                pm = torch.randperm(m, generator=g).to(dev)
                pn = torch.randperm(n, generator=g).to(dev)
                W = (P[pm] * sig) @ Q[pn].T
                if kind == "lowrank_noise":
                    W = W + 1e-4 * float(sig[0]) / max(m, n) ** 0.5 * torch.randn(m, n, generator=g, dtype=torch.float64).to(dev)
                W = W * (0.02 * (m * n) ** 0.5 / float(W.norm()))
                mats.append(W.float().contiguous())
                stats.append(F.make_stat(stat, n, seed=b).to(dev))
        scs = ops.make_scale_batched(stats, None, alpha)
        torch.cuda.synchronize()
        gen_s = time.time() - t0
        ops.svd_batched(mats[:4], scs[:4])   # warm-up (streams, attributes)
        torch.cuda.synchronize()
        t0 = time.time()
        U, S, V, infos = ops.svd_batched(mats, scs)
        torch.cuda.synchronize()
        dt = time.time() - t0
        rec = {"family": kind, "stat": stat, "alpha": alpha, "svd_per_s": round(B / dt, 2), "ms_per_call": round(dt * 1e3, 1),
               "sweeps_min": min(i.sweeps for i in infos), "sweeps_max": max(i.sweeps for i in infos),
               "status_max": max(i.status for i in infos), "reduced": infos[0].reduced, "reduce_fallback": infos[0].reduce_fallback, "gram_retry": any(i.gram_retry for i in infos),
               "plain_retry": infos[0].plain_retry, "split": infos[0].split, "gen_s": round(gen_s, 1)}
        if not a.no_check:
            for b in (0, B - 1):
                Ws = (mats[b].double() * scs[b].double().unsqueeze(0))
                So = torch.linalg.svdvals(Ws.float().double())  # the fp32 product the oracle factorises, spectrum in fp64
                r = int(0.9 * m * n / (m + n))
                r = k // 4 if kind == "lowrank_noise" else r
                rec[f"sigma_rel_err_top_r_b{b}"] = float(((S[b].double() - So).abs() / So)[:r].max())
                rec[f"sigma_abs_err_b{b}"] = float((S[b].double() - So).abs().max() / So[0])
            rec["cond_top_r"] = float(So[0] / So[r - 1])
        del U, V
        ops.svd_profile(True)
        _, _, _, infos2 = ops.svd_batched(mats, scs)
        prof = ops.svd_profile()
        ops.svd_profile(False)
        rec["unsplit_sweep_ms"] = [round(x, 1) for x in prof["sweep_ms"]]
        rec["unsplit_sweep_rotated"] = prof["sweep_rotated"]
        rec["unsplit_class_ms"] = {c: round(prof[c]["ms"], 1) for c in ("pack", "evd", "supgram", "snapshot", "finalize", "gram1", "update1", "sgram", "supdate")}
        print(json.dumps(rec), flush=True)
        del mats, scs, S
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
