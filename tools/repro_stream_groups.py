"""Reproducer / probe for DESIGN.md 3.8 (b): "several stream groups => stale reads between kernels of the same stream".

Runs the SAME batch of 4096 x 4096 problems through asvd_svd_batched under different launch structures and compares every run with
the one-stream-group result (the kernels are deterministic: any difference in S / U / V bits or in the sweep count is a hazard hit):

    groups=1                     baseline (what the library does by default)
    groups=G fence=1             the round-2 remedy: agent-scope acquire at kernel start + release at kernel end
    groups=G fence=0             the hazardous configuration, repeated --reps times

Prints one JSON line per configuration: mismatching problems, sweep counts, wall time.  Environment knobs are read by the library at
call time (ASVD_GROUPS, ASVD_FENCE), so one process covers all configurations.  Usage: python tools/repro_stream_groups.py [--batch 16]
[--groups 2] [--reps 6] [--n 4096]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--n", type=int, default=4096)
    args = ap.parse_args()
    import torch
    from asvd4llm_amd import ops
    from bench import synth
    dev = torch.device("cuda", 0)
    mats = [synth(args.n, args.n, seed=500 + b)[0].to(dev) for b in range(args.batch)]

    def run(groups, fence):
        os.environ["ASVD_GROUPS"] = str(groups)
        if fence is None:
            os.environ.pop("ASVD_FENCE", None)
        else:
            os.environ["ASVD_FENCE"] = str(fence)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        U, S, V, infos = ops.svd_batched(mats)
        torch.cuda.synchronize()
        return (U, S, V, infos), time.perf_counter() - t0

    def diff(res, ref):
        bad = []
        for b in range(args.batch):
            same = torch.equal(res[1][b], ref[1][b]) and torch.equal(res[0][b], ref[0][b]) and torch.equal(res[2][b], ref[2][b])
            if not same or res[3][b].sweeps != ref[3][b].sweeps:
                rel = ((res[1][b].double() - ref[1][b].double()).abs().max() / ref[1][b].double().max()).item()
                bad.append({"problem": b, "sweeps": res[3][b].sweeps, "ref_sweeps": ref[3][b].sweeps, "status": res[3][b].status, "max_rel_dS": rel})
        return bad

    one, t_one = run(1, None)
    one2, t_one2 = run(1, None)
    print(json.dumps({"config": "groups=1 (run twice: determinism of the default)", "seconds": t_one2, "first_seconds": t_one,
                      "n_mismatch": len(diff(one2, one)), "sweeps": [i.sweeps for i in one[3]]}), flush=True)
    # the row splits of the streaming kernels depend on the problems per launch, so a grouped run legitimately differs from the
    # one-group run in the last bits: the reference for the hazardous configuration is the SAME launch structure with fences on
    ref, t_ref = run(args.groups, 1)
    ref2, t_ref2 = run(args.groups, 1)
    print(json.dumps({"config": f"groups={args.groups} fence=1 (run twice)", "seconds": t_ref2, "n_mismatch": len(diff(ref2, ref)),
                      "sweeps": [i.sweeps for i in ref[3]],
                      "max_rel_dS_vs_one_group": max(((ref[1][b].double() - one[1][b].double()).abs().max() / one[1][b].double().max()).item() for b in range(args.batch))}),
          flush=True)
    for rep in range(args.reps):
        res, t = run(args.groups, 0)
        bad = diff(res, ref)
        print(json.dumps({"config": f"groups={args.groups} fence=0", "rep": rep, "seconds": t, "n_mismatch": len(bad), "mismatches": bad[:6],
                          "sweeps": [i.sweeps for i in res[3]]}), flush=True)
    os.environ["ASVD_GROUPS"] = "1"
    os.environ.pop("ASVD_FENCE", None)


if __name__ == "__main__":
    main()
