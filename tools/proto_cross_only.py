"""Convergence experiment: block one-sided Jacobi where the solve of a panel pair rotates (a) all pairs of its 64 columns in the odd-even
transposition order (what evd_wave.hip does today) or (b) only the 1024 cross pairs, in the ring order (S columns move right, T columns move left,
positions interleaved)."""
import numpy as np, sys, time

def rot_pairs(X, V, I, J, thr):
    # rotate disjoint column pairs (I[k], J[k]) of X (one-sided), accumulate nothing else; returns max |cos| seen
    a = np.einsum('ij,ij->j', X[:, I], X[:, I]); d = np.einsum('ij,ij->j', X[:, J], X[:, J]); b = np.einsum('ij,ij->j', X[:, I], X[:, J])
    cosv = np.abs(b) / np.sqrt(np.maximum(a * d, 1e-300))
    act = cosv > thr
    zeta = (d - a) / (2 * np.where(b == 0, 1e-300, b))
    t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta)); t = np.where(act, t, 0.0)
    c = 1 / np.sqrt(1 + t * t); s = t * c
    xi, xj = X[:, I].copy(), X[:, J].copy()
    X[:, I] = c * xi - s * xj
    X[:, J] = s * xi + c * xj
    return cosv.max() if len(cosv) else 0.0, int(act.sum())

def solve_full(X, cols, thr):
    # odd-even transposition with swap over the positions of `cols` (64): every pair once
    pos = list(cols); n = len(pos); mx = 0.0; nr = 0
    for ph in range(n):
        st = ph & 1
        I = [pos[k] for k in range(st, n - 1, 2)]; J = [pos[k + 1] for k in range(st, n - 1, 2)]
        m, r = rot_pairs(X, None, np.array(I), np.array(J), thr); mx = max(mx, m); nr += r
        for k in range(st, n - 1, 2): pos[k], pos[k + 1] = pos[k + 1], pos[k]
    return mx, nr

def solve_cross(X, S, T, thr):
    # ring of 2w positions, S at even, T at odd; phase A pairs (2k, 2k+1), phase B pairs (2k+1, 2k+2 mod n), swap after each: w phases = all cross pairs
    w = len(S); n = 2 * w; pos = [0] * n
    for i in range(w): pos[2 * i] = S[i]; pos[2 * i + 1] = T[i]
    mx = 0.0; nr = 0
    for ph in range(w):
        st = ph & 1
        ks = list(range(st, n, 2))
        I = [pos[k] for k in ks]; J = [pos[(k + 1) % n] for k in ks]
        m, r = rot_pairs(X, None, np.array(I), np.array(J), thr); mx = max(mx, m); nr += r
        for k in ks: pos[k], pos[(k + 1) % n] = pos[(k + 1) % n], pos[k]
    return mx, nr

def run(X, w, mode, tol=1e-7, max_sweeps=20):
    n = X.shape[1]; P = n // w
    X = X.copy(); out = []
    for sw in range(max_sweeps):
        mx = 0.0; nr = 0
        # internal step: panels (2k, 2k+1) full solve
        for k in range(0, P, 2):
            m, r = solve_full(X, list(range(k * w, (k + 2) * w)), tol); mx = max(mx, m); nr += r
        # super-steps: super-panels of 2 panels; XOR distance D over super-panels; sub-pairs (0,2)(1,3) then (0,3)(1,2)
        ns = P // 2
        for D in range(1, ns):
            for a in range(ns):
                b = a ^ D
                if b < a: continue
                pa = [2 * a, 2 * a + 1]; pb = [2 * b, 2 * b + 1]
                for (x, y) in ((pa[0], pb[0]), (pa[1], pb[1]), (pa[0], pb[1]), (pa[1], pb[0])):
                    S = list(range(x * w, (x + 1) * w)); T = list(range(y * w, (y + 1) * w))
                    if mode == 'full': m, r = solve_full(X, S + T, tol)
                    else: m, r = solve_cross(X, S, T, tol)
                    mx = max(mx, m); nr += r
        out.append((mx, nr))
        print(mode, 'sweep', sw + 1, 'max|cos| %.2e rotations %d' % (mx, nr), flush=True)
        if mx < tol: break
    return out

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, n)) * (rng.random(n) ** 2 * 3 + 0.05)[None, :]   # activation-like column scales
    # precondition as the pipeline does: sort columns by norm, QR, Jacobi on R^T
    order = np.argsort(-np.linalg.norm(A, axis=0)); R = np.linalg.qr(A[:, order])[1]
    X0 = R.T.copy()
    for mode in ('full', 'cross'):
        t0 = time.time(); run(X0, w, mode); print(mode, 'time', round(time.time() - t0, 1))
