"""CPU prototype behind DESIGN.md §7 'Pair ordering': sweeps of the fp32 block-Jacobi emulator (tools/proto_block_jacobi.py) on the
Cholesky-preconditioned, norm-sorted matrix under different panel-pair schedules.

  python tools/proto_pair_order.py 1024 rr        round-robin tournament (8 sweeps at n = 1024)
  python tools/proto_pair_order.py 1024 nat       XOR schedule d = 1, 2, ..., P-1 (6)
  python tools/proto_pair_order.py 1024 rev|pop|hib  other permutations of the XOR distances (6-7)
  python tools/proto_pair_order.py 1024 rowcyc    serial row-cyclic order, the non-parallel reference (6)
  python tools/proto_pair_order.py 1024 dup7      local levels d = 1..7 twice per sweep (half a sweep ahead)
"""
import sys
import time

import numpy as np

sys.path.insert(0, "tools")
import proto_block_jacobi as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2] if len(sys.argv) > 2 else "nat"
W, s = P.llm_like(n, n)
Ws = (W * s[None, :]).astype(np.float32)
S64 = np.linalg.svd(Ws.astype(np.float64), compute_uv=False)
r = int(n * n * 0.9) // (2 * n)
order = np.argsort(-np.linalg.norm(Ws, axis=0))
Q, R = np.linalg.qr(Ws[:, order].astype(np.float64))
A = R.T.copy()
nb = n // 32
ORIG = P.rr_pairs


def xor_pairs(nb_, d):
    return [(i, i ^ d) for i in range(nb_) if i < (i ^ d)]


def sequence():
    ds = list(range(1, nb))
    if which == "rev":
        return ds[::-1]
    if which == "pop":
        return sorted(ds, key=lambda d: (bin(d).count("1"), d))
    if which == "hib":
        return sorted(ds, key=lambda d: (-d.bit_length(), d))
    return ds


D = sequence()


def pairs(nb_, st):
    if nb_ != nb or which == "rr":
        return ORIG(nb_, st)  # the inner 64x64 eigen-solve keeps its own ordering
    if which == "rowcyc":
        return [(st, j) for j in range(st + 1, nb_)]
    if which == "dup7":
        if st == 0:
            return [p for _ in range(2) for d in range(1, 8) for p in xor_pairs(nb_, d)]
        return [] if st < 7 else xor_pairs(nb_, st + 1)
    return xor_pairs(nb_, D[st])


P.rr_pairs = pairs
print("==", which, n, flush=True)
t0 = time.time()
P.block_jacobi_svd(A.astype(np.float32), B=32, inner_sweeps=2, tol=1e-6, sigma_true=S64, r=r, accumulate_v=False)
print("time", time.time() - t0, flush=True)
