"""Throw-away numerical prototype (CPU, numpy fp32) of the block one-sided Jacobi SVD that the HIP
kernels implement.  Used to size block width / inner sweeps / tolerances before writing device code.
Not part of the product path, not imported by it."""
import sys, time
import numpy as np

def llm_like(m, n, seed=233, n_calib=32):
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((m, n)) * 0.02).astype(np.float32)
    k = max(1, int(0.005 * n)); oc = rng.choice(n, k, replace=False); W[:, oc] *= 20
    scal = (n_calib * np.abs(rng.standard_normal(n))).astype(np.float32)
    k = max(1, int(0.01 * n)); oc = rng.choice(n, k, replace=False); scal[oc] *= 30
    scal = scal.astype(np.float16)
    s = (scal.astype(np.float16) ** np.float16(0.5) + np.float16(1e-6)).astype(np.float32)
    return W, s

def rr_pairs(nb, step):
    """round-robin tournament: nb even, step in [0, nb-1): list of (i,j)"""
    idx = [0] + [1 + (k + step) % (nb - 1) for k in range(nb - 1)]
    return [(min(idx[k], idx[nb - 1 - k]), max(idx[k], idx[nb - 1 - k])) for k in range(nb // 2)]

def evd_jacobi(G, max_sweeps, tol):
    """two-sided Jacobi on symmetric G (fp32 storage), parallel round-robin ordering, rotation scalars in fp64.
    returns Q (fp32), number of sweeps, initial max scaled offdiag"""
    k = G.shape[0]
    G = G.copy(); Q = np.eye(k, dtype=np.float32)
    d = np.sqrt(np.maximum(np.diag(G), 1e-300).astype(np.float64))
    off0 = np.max(np.abs(G - np.diag(np.diag(G))) / np.outer(d, d))
    nsw = 0
    for sw in range(max_sweeps):
        d = np.sqrt(np.maximum(np.diag(G).astype(np.float64), 1e-300))
        off = np.max(np.abs(G - np.diag(np.diag(G))) / np.outer(d, d))
        if off < tol: break
        nsw += 1
        for st in range(k - 1):
            pr = rr_pairs(k, st)
            p = np.array([a for a, b in pr]); q = np.array([b for a, b in pr])
            gpp = G[p, p].astype(np.float64); gqq = G[q, q].astype(np.float64); gpq = G[p, q].astype(np.float64)
            rot = np.abs(gpq) > tol * 0.1 * np.sqrt(np.abs(gpp * gqq))
            zeta = np.where(rot, (gqq - gpp) / (2 * np.where(gpq == 0, 1, gpq)), 0)
            t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta)); t = np.where(zeta == 0, 1.0, t)
            t = np.where(rot, t, 0.0)
            c = 1 / np.sqrt(1 + t * t); s = c * t
            c = c.astype(np.float32); s = s.astype(np.float32)
            # columns
            Gp = G[:, p].copy(); Gq = G[:, q].copy()
            G[:, p] = c * Gp - s * Gq; G[:, q] = s * Gp + c * Gq
            Gp = G[p, :].copy(); Gq = G[q, :].copy()
            G[p, :] = c[:, None] * Gp - s[:, None] * Gq; G[q, :] = s[:, None] * Gp + c[:, None] * Gq
            Qp = Q[:, p].copy(); Qq = Q[:, q].copy()
            Q[:, p] = c * Qp - s * Qq; Q[:, q] = s * Qp + c * Qq
    if nsw > 0:
        order = np.argsort(-np.diag(G), kind='stable'); Q = Q[:, order]
    return Q, nsw, off0

def block_jacobi_svd(A, B=32, inner_sweeps=3, tol=1e-6, max_sweeps=16, accumulate_v=True, exact_inner=False, log=print,
                     sigma_true=None, r=None):
    A = A.astype(np.float32).copy(); m, n = A.shape
    assert n % (2 * B) == 0
    nb = n // B
    V = np.eye(n, dtype=np.float32) if accumulate_v else None
    for sweep in range(max_sweeps):
        maxoff = 0.0; nrot = 0; ninner = 0
        for st in range(nb - 1):
            for (I, J) in rr_pairs(nb, st):
                cols = np.r_[I * B:(I + 1) * B, J * B:(J + 1) * B]
                P = A[:, cols]
                G = (P.T @ P).astype(np.float32)
                if exact_inner:
                    d = np.sqrt(np.maximum(np.diag(G).astype(np.float64), 1e-300))
                    off0 = np.max(np.abs(G - np.diag(np.diag(G))) / np.outer(d, d))
                    if off0 >= tol:
                        w, Q = np.linalg.eigh(G.astype(np.float64)); Q = Q[:, ::-1].astype(np.float32); nsw = 1
                    else: nsw = 0
                else:
                    Q, nsw, off0 = evd_jacobi(G, inner_sweeps, tol)
                maxoff = max(maxoff, off0)
                if nsw > 0:
                    nrot += 1; ninner += nsw
                    A[:, cols] = P @ Q
                    if accumulate_v: V[:, cols] = V[:, cols] @ Q
        sig = np.linalg.norm(A.astype(np.float64), axis=0)
        if accumulate_v: sigc = sig / np.linalg.norm(V.astype(np.float64), axis=0)
        else: sigc = sig
        msg = f"sweep {sweep+1}: maxoff(start)={maxoff:.3e} active_pairs={nrot}/{(nb-1)*nb//2} inner_sweeps={ninner}"
        if sigma_true is not None:
            ss = np.sort(sigc)[::-1]; rr = r or n
            e = np.max(np.abs(ss[:rr] - sigma_true[:rr]) / sigma_true[:rr])
            ss2 = np.sort(sig)[::-1]
            e2 = np.max(np.abs(ss2[:rr] - sigma_true[:rr]) / sigma_true[:rr])
            msg += f" top-r sigma relerr corrected={e:.2e} raw={e2:.2e}"
        log(msg)
        if maxoff < tol: break
    return A, V, sweep + 1

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    inner = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-6
    exact = (len(sys.argv) > 5 and sys.argv[5] == "exact")
    W, s = llm_like(n, n)
    Ws = W * s[None, :]
    t0 = time.time(); S64 = np.linalg.svd(Ws.astype(np.float64), compute_uv=False); print("fp64 svd", time.time() - t0)
    r = int(n * n * 0.9) // (2 * n)
    S32 = np.linalg.svd(Ws, compute_uv=False)
    print("cond", S64[0] / S64[-1], "lapack fp32 top-r err", np.max(np.abs(S32[:r] - S64[:r]) / S64[:r]))
    t0 = time.time()
    A, V, nsw = block_jacobi_svd(Ws, B=B, inner_sweeps=inner, tol=tol, sigma_true=S64, r=r, exact_inner=exact)
    print("time", time.time() - t0)
