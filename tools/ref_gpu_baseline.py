"""Context only — never imported by the product: what the reference LITERALLY runs, timed on the same MI355X through PyTorch-ROCm.

The reference loads the model with device_map="auto" (asvd.py:25-27) and calls `torch.svd_lowrank(w, q=rank)` on the GPU tensor
(modules/svd_linear.py:65): a randomized range finder (5 GEMMs + 5 thin QRs + one small exact SVD, torch/_lowrank.py) executed by
rocBLAS / rocSOLVER.  The parity oracle named by BASELINE.json is the exact `torch.linalg.svd`, timed here on the device as well
(rocSOLVER gesvd/gesvdj behind torch).  Same synthetic matrices as bench.py.  Prints one JSON line.

    python tools/ref_gpu_baseline.py [--m 4096 --n 4096 --reps 3 --skip_exact]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--skip_exact", action="store_true", help="skip torch.linalg.svd on the device (rocSOLVER can take minutes at 4096^2)")
    ap.add_argument("--exact_timeout_s", type=float, default=300.0)
    args = ap.parse_args()
    import torch
    from bench import synth
    dev = torch.device("cuda", 0)
    W, scal = synth(args.m, args.n, seed=233)
    s = scal.float().pow(0.5) + 1e-6
    Ws = (W * s.view(1, -1)).to(dev)
    torch.cuda.synchronize()

    def timed(fn, reps):
        fn()  # warm-up (library init, workspace allocation)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], ts

    out = {"what": "reference calls on the MI355X through PyTorch-ROCm (context only; not the target, not used by the product)",
           "shape": [args.m, args.n], "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}
    for q in (512, int(args.m * args.n * 0.9) // (args.m + args.n)):
        torch.manual_seed(233)
        med, ts = timed(lambda: torch.svd_lowrank(Ws, q=q), args.reps)
        out[f"svd_lowrank_q{q}_s"] = med
        out[f"svd_lowrank_q{q}_all_s"] = ts
    if not args.skip_exact:
        t0 = time.perf_counter()
        try:
            med, ts = timed(lambda: torch.linalg.svd(Ws, full_matrices=False), 1 if args.m * args.n >= 4096 * 4096 else args.reps)
            out["linalg_svd_s"] = med
            out["linalg_svd_all_s"] = ts
        except Exception as e:  # noqa: BLE001
            out["linalg_svd_error"] = repr(e)
        out["linalg_svd_leg_wall_s"] = time.perf_counter() - t0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
