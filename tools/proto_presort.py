import sys, time
import numpy as np
sys.path.insert(0, "tools")
from proto_block_jacobi import llm_like, block_jacobi_svd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, s = llm_like(n, n); Ws = (W * s[None, :]).astype(np.float32)
S64 = np.linalg.svd(Ws.astype(np.float64), compute_uv=False); r = int(n * n * 0.9) // (2 * n)
def run(tag, A):
    print("==", tag, flush=True); t0 = time.time()
    block_jacobi_svd(A.astype(np.float32), B=32, inner_sweeps=2, tol=1e-6, sigma_true=S64, r=r, accumulate_v=False)
    print("time", time.time() - t0, flush=True)
order = np.argsort(-np.linalg.norm(Ws, axis=0))
run("presorted desc", Ws[:, order])
run("presorted asc", Ws[:, order[::-1]])
run("transposed (rows)", Ws.T.copy())
