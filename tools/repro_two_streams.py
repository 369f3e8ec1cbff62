"""Probe for the concurrency hazard of DESIGN.md 3.8: two host threads, each with its own stream and workspace, run asvd_svd_batched on
their own problems; every result is compared bit for bit (and sweep for sweep) with the same call made alone.  Variants (one JSON line
each) separate the candidate causes:
    serial x2                 determinism of the serial runs themselves
    threads                   the failing configuration of tests/test_gpu_concurrency.py (no third stream)
    threads+gemm              plus a foreign stream of torch GEMMs
    one+gemm                  ONE svd thread next to the GEMM stream
    threads fence=both|acq|rel   agent-scope fences inside the kernels (ASVD_FENCE=1|2|3)
    threads workfill=0|255    workspace zero- / NaN-filled before every call (reads of unwritten workspace would show here)
Usage: python tools/repro_two_streams.py [--nprob 8] [--n 4096] [--rounds 3]"""
import argparse
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nprob", type=int, default=8)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--extra", type=str, action="append", default=[], help="additional two-thread variant NAME:K=V[,K=V...] (environment knobs of the library)")
    args = ap.parse_args()
    import torch
    from asvd4llm_amd import ops
    from bench import synth
    dev = torch.device("cuda", 0)
    jobs = [[synth(args.n, args.n, seed=900 + 100 * j + b)[0].to(dev) for b in range(args.nprob)] for j in range(2)]

    def same(a, b):
        bits = all(torch.equal(x, y) for p, q in zip(a[:3], b[:3]) for x, y in zip(p, q))
        return bits, [i.sweeps for i in a[3]], max(((x.double() - y.double()).abs().max() / y.double().max()).item() for x, y in zip(a[1], b[1]))

    def concurrent(njobs, gemm):
        results = [[] for _ in range(njobs)]
        stop = threading.Event()
        bar = threading.Barrier(njobs + (1 if gemm else 0))

        def w(i):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                bar.wait()
                for _ in range(args.rounds):
                    results[i].append(ops.svd_batched(jobs[i]))
            st.synchronize()

        def g():
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
                b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
                bar.wait()
                while not stop.is_set():
                    for _ in range(8):
                        a = ((a @ b) * 1e-2).clamp_(-1, 1)
                    st.synchronize()

        ts = [threading.Thread(target=w, args=(i,)) for i in range(njobs)]
        tg = threading.Thread(target=g) if gemm else None
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
        return results

    def setenv(fence=None, fill=None):
        for k, v in (("ASVD_FENCE", fence), ("ASVD_DEBUG_WORKFILL", fill)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def report(name, refs, res):
        out = {"config": name, "runs": []}
        nbad = 0
        for i, rr in enumerate(res):
            for k, r in enumerate(rr):
                bits, sw, rel = same(r, refs[i])
                nbad += 0 if (bits and sw == [x.sweeps for x in refs[i][3]]) else 1
                out["runs"].append({"job": i, "round": k, "bit_identical": bits, "sweeps": sw, "max_rel_dS": rel})
        out["n_runs"] = sum(len(r) for r in res)
        out["n_differ"] = nbad
        out["ref_sweeps"] = [[x.sweeps for x in r[3]] for r in refs]
        print(json.dumps(out), flush=True)

    variants = [("threads", 2, False, None, None), ("threads+gemm", 2, True, None, None), ("one+gemm", 1, True, None, None),
                ("threads fence=both", 2, False, 1, None), ("threads fence=acq", 2, False, 2, None), ("threads fence=rel", 2, False, 3, None),
                ("threads workfill=0", 2, False, None, 0), ("threads workfill=255", 2, False, None, 255)]
    variants = [v + ({},) for v in variants]
    for ex in args.extra:
        nm, kv = ex.split(":", 1)
        variants.append(("threads " + nm, 2, False, None, None, dict(x.split("=", 1) for x in kv.split(","))))
    for name, nj, gemm, fence, fill, extra_env in variants:
        if args.only and not any(o in name for o in args.only.split("|")):
            continue
        setenv(fence, fill)
        for k, v in extra_env.items():
            os.environ[k] = v
        refs = [ops.svd_batched(j) for j in jobs[:nj]]
        torch.cuda.synchronize()
        refs2 = [ops.svd_batched(j) for j in jobs[:nj]]
        torch.cuda.synchronize()
        if name == "threads":
            report("serial x2", refs, [[r] for r in refs2])
        report(name, refs, concurrent(nj, gemm))
        for k in extra_env:
            os.environ.pop(k, None)
    setenv()


if __name__ == "__main__":
    main()
