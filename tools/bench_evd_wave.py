"""Microbenchmark of the wave-local 64x64 eigen-solver through its test hook: time per inner sweep (64 phases) as a function of the number
of waves in flight.  1024 waves = one per SIMD of the 256 CUs.  Prints one JSON line per batch size."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from asvd4llm_amd import _lib
    lib = _lib.load(True)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((256, 64)).astype(np.float32)
    G1 = torch.from_numpy((X.T @ X).astype(np.float32))
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for B in (256, 1024, 2048, 3072, 4096, 8192):
        G = G1.unsqueeze(0).repeat(B, 1, 1).contiguous().to(dev)
        Q = torch.empty_like(G)
        Go = torch.empty_like(G)
        d = torch.empty((B, 64), dtype=torch.float32, device=dev)
        c = torch.empty((B, 64), dtype=torch.float32, device=dev)
        r = torch.empty((B, 64), dtype=torch.int32, device=dev)
        m = torch.empty((B, 2), dtype=torch.float32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run(sw, reps=5):
            lib.asvd_test_evd_wave(p(G), B, sw, p(Q), p(d), p(r), p(c), p(Go), p(m), st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.asvd_test_evd_wave(p(G), B, sw, p(Q), p(d), p(r), p(c), p(Go), p(m), st)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        t1, t3 = run(1), run(3)
        per_sweep = (t3 - t1) / 2
        print(json.dumps({"waves": B, "t_1sweep_us": 1e6 * t1, "t_3sweeps_us": 1e6 * t3, "us_per_sweep": 1e6 * per_sweep,
                          "us_per_phase": 1e6 * per_sweep / 64, "solves_per_us": B / (1e6 * per_sweep)}), flush=True)


if __name__ == "__main__":
    main()
