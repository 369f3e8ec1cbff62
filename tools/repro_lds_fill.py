"""Which LDS region does the LDS eigen-solver (evd_kernel, ASVD_EVDW=0) read before writing it?  Serial runs, one problem batch: the
reference result (LDS zero-filled at kernel entry, ASVD_FENCE=4) against runs whose LDS regions are pre-filled with NaN one at a time
(ASVD_EVD_LDSFILL=1<<r; regions G | Qs | sdiag | sb | redmax | cscale | rnk).  A region whose NaN fill changes the result is read
before it is written.  One JSON line per region."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from asvd4llm_amd import ops
    from bench import synth
    dev = torch.device("cuda", 0)
    os.environ["ASVD_EVDW"] = "0"
    for n, nprob in ((1024, 4), (256, 4)):
        mats = [synth(n, n, seed=40 + b)[0].to(dev) for b in range(nprob)]
        os.environ["ASVD_FENCE"] = "4"
        os.environ.pop("ASVD_EVD_LDSFILL", None)
        ref = ops.svd_batched(mats)
        torch.cuda.synchronize()
        os.environ.pop("ASVD_FENCE")
        plain = ops.svd_batched(mats)
        print(json.dumps({"n": n, "config": "no fill vs zero fill", "bit_identical": all(torch.equal(a, b) for a, b in zip(plain[1], ref[1])),
                          "sweeps": [i.sweeps for i in plain[3]], "ref_sweeps": [i.sweeps for i in ref[3]]}), flush=True)
        for r, name in enumerate(["G", "Qs", "sdiag", "sb", "redmax", "cscale", "rnk"]):
            os.environ["ASVD_EVD_LDSFILL"] = str(1 << r)
            res = ops.svd_batched(mats)
            torch.cuda.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(res[1], ref[1]))
            nan = any(bool(torch.isnan(a).any()) for a in res[1])
            print(json.dumps({"n": n, "region": name, "bit_identical_to_zero_fill": same, "nan_in_S": nan, "status": [i.status for i in res[3]],
                              "sweeps": [i.sweeps for i in res[3]]}), flush=True)
        os.environ.pop("ASVD_EVD_LDSFILL", None)


if __name__ == "__main__":
    main()
