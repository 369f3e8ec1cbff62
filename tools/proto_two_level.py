"""CPU prototype (numpy fp32) of the TWO-LEVEL block Jacobi sweep sized for round 2: 64-column super-panels (two
32-column panels), XOR schedule over super-panels, per super-pair ONE cross-Gram pass, an inner block-Jacobi on the 128x128 Gram
matrix (64x64 sub-solves on the four 32-blocks, Gram updated two-sidedly, Q accumulated), ONE 128-wide update pass.

  python tools/proto_two_level.py 1024 single        single-level XOR schedule, 32-wide panels (round-1 scheme)
  python tools/proto_two_level.py 1024 two           two-level: internal step once per sweep, then cross inner steps per super-pair
  python tools/proto_two_level.py 1024 two_full      two-level with the internal sub-pairs re-solved inside every super-pair
"""
import sys
import time

import numpy as np

sys.path.insert(0, "tools")
import proto_block_jacobi as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2] if len(sys.argv) > 2 else "two"
inner_sw = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = 32
W, s = P.llm_like(n, n)
Ws = (W * s[None, :]).astype(np.float32)
S64 = np.linalg.svd(Ws.astype(np.float64), compute_uv=False)
r = int(n * n * 0.9) // (2 * n)
order = np.argsort(-np.linalg.norm(Ws, axis=0))
Qf, R = np.linalg.qr(Ws[:, order].astype(np.float64))
A = R.T.astype(np.float32).copy()
nb = n // B
nsolves = 0


def solve(G, tol=1e-6):
    global nsolves
    nsolves += 1
    return P.evd_jacobi(G, inner_sw, tol)


def pair_step(A, I, J, stats):
    cols = np.r_[I * B:(I + 1) * B, J * B:(J + 1) * B]
    Pn = A[:, cols]
    G = (Pn.T @ Pn).astype(np.float32)
    Q, nsw, off0 = solve(G)
    stats[0] = max(stats[0], off0)
    if nsw > 0:
        A[:, cols] = Pn @ Q
        stats[1] += 1


def sweep_single(A, stats, dup=0):
    ds = list(range(1, dup + 1)) + list(range(1, nb))
    for d in ds:
        for i in range(nb):
            if i < (i ^ d):
                pair_step(A, i, i ^ d, stats)


NB = int(sys.argv[4]) if len(sys.argv) > 4 else 2  # panels per super-panel
GLOBAL_SORT = not (len(sys.argv) > 5 and sys.argv[5] == "nosort")


def super_pair(A, S, T, stats, full):
    blocks = [NB * S + i for i in range(NB)] + [NB * T + i for i in range(NB)]
    cols = np.concatenate([np.arange(b * B, (b + 1) * B) for b in blocks])
    Pn = A[:, cols]
    G = (Pn.T @ Pn).astype(np.float32)  # device: carried diagonal blocks + one cross-Gram pass
    Qacc = np.eye(2 * NB * B, dtype=np.float32)
    steps = [[(a, NB + (a + t) % NB) for a in range(NB)] for t in range(NB)]
    if full:
        steps.append([(0, 1), (2, 3)])
    rotated = False
    for st in steps:
        for (a, b) in st:
            idx = np.r_[a * B:(a + 1) * B, b * B:(b + 1) * B]
            Q, nsw, off0 = solve(G[np.ix_(idx, idx)])
            stats[0] = max(stats[0], off0)
            if nsw > 0:
                rotated = True
                G[:, idx] = G[:, idx] @ Q
                G[idx, :] = Q.T @ G[idx, :]
                Qacc[:, idx] = Qacc[:, idx] @ Q
    if rotated:
        perm = np.argsort(-np.diag(G), kind="stable") if GLOBAL_SORT else np.arange(G.shape[0])
        A[:, cols] = Pn @ Qacc[:, perm]
        stats[1] += 1


def sweep_two(A, stats, full):
    ps = nb // NB
    for d in range(1, NB):  # internal pairs (the d < NB steps of the single-level schedule)
        for i in range(nb):
            if i < (i ^ d):
                pair_step(A, i, i ^ d, stats)
    for D in range(1, ps):
        for S in range(ps):
            if S < (S ^ D):
                super_pair(A, S, S ^ D, stats, full)


print("==", which, n, "inner sweeps", inner_sw, flush=True)
t0 = time.time()
for sweep in range(14):
    stats = [0.0, 0]
    nsolves = 0
    if which == "single":
        sweep_single(A, stats)
    elif which == "single_dup":
        sweep_single(A, stats, dup=min(7, nb // 16 - 1))
    else:
        sweep_two(A, stats, which == "two_full")
    sig = np.sort(np.linalg.norm(A.astype(np.float64), axis=0))[::-1]
    e = np.max(np.abs(sig[:r] - S64[:r]) / S64[:r])
    print(f"sweep {sweep+1}: maxoff(start)={stats[0]:.3e} rotated={stats[1]} solves={nsolves} top-r sigma relerr={e:.2e}  t={time.time()-t0:.0f}s", flush=True)
    if stats[0] < 1e-6 or (sweep > 3 and e < 2e-6 and stats[0] < 1e-3):
        break
