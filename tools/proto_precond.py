"""Throw-away CPU experiment: does QR preconditioning (Drmac-Veselic: Jacobi on R^T after a QR with norm-sorted / pivoted columns)
cut the number of block-Jacobi sweeps?  Uses the fp32 block-Jacobi prototype."""
import sys, time
import numpy as np, scipy.linalg as sl
sys.path.insert(0, "tools")
from proto_block_jacobi import llm_like, block_jacobi_svd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, s = llm_like(n, n); Ws = (W * s[None, :]).astype(np.float32)
S64 = np.linalg.svd(Ws.astype(np.float64), compute_uv=False); r = int(n * n * 0.9) // (2 * n)
def run(tag, A):
    print("==", tag, flush=True)
    t0 = time.time()
    block_jacobi_svd(A.astype(np.float32), B=32, inner_sweeps=2, tol=1e-6, sigma_true=S64, r=r, accumulate_v=False)
    print("time", time.time() - t0, flush=True)
run("plain", Ws)
# sorted columns + unpivoted QR, Jacobi on R^T
order = np.argsort(-np.linalg.norm(Ws, axis=0))
Q, R = np.linalg.qr(Ws[:, order].astype(np.float64))
run("sorted QR, jacobi on R^T (=L)", R.T.copy())
run("sorted QR, jacobi on R", R.copy())
Q2, R2, piv = sl.qr(Ws.astype(np.float64), pivoting=True, mode="economic")
run("pivoted QR, jacobi on R^T", R2.T.copy())
# second QR (R^T = Q2 R2 -> jacobi on R2^T)
Q3, R3 = np.linalg.qr(R.T)
run("sorted QR twice, jacobi on R3^T", R3.T.copy())
