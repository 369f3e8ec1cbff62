import sys, time
import numpy as np
sys.path.insert(0, "tools")
from proto_block_jacobi import llm_like, block_jacobi_svd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
which = sys.argv[2] if len(sys.argv) > 2 else "plain"
W, s = llm_like(n, n); Ws = (W * s[None, :]).astype(np.float32)
S64 = np.linalg.svd(Ws.astype(np.float64), compute_uv=False); r = int(n * n * 0.9) // (2 * n)
if which == "plain":
    A = Ws
elif which == "qr_rt":   # unpivoted QR of the column-sorted matrix, Jacobi on R^T
    order = np.argsort(-np.linalg.norm(Ws, axis=0))
    Q, R = np.linalg.qr(Ws[:, order].astype(np.float64)); A = R.T.copy()
elif which == "qr_rt_nosort":
    Q, R = np.linalg.qr(Ws.astype(np.float64)); A = R.T.copy()
elif which == "qr2_rt":
    order = np.argsort(-np.linalg.norm(Ws, axis=0))
    Q, R = np.linalg.qr(Ws[:, order].astype(np.float64)); Q2, R2 = np.linalg.qr(R.T); A = R2.T.copy()
print("==", which, n, flush=True); t0 = time.time()
block_jacobi_svd(A.astype(np.float32), B=32, inner_sweeps=2, tol=1e-6, sigma_true=S64, r=r, accumulate_v=False)
print("time", time.time() - t0, flush=True)
