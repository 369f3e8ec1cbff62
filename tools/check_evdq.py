"""Cooperative (four waves per solve) vs wave-local eigen-solver launches: bit-identical SVD results and wall time.
usage: check_evdq.py run <evdq 0|1> <n> <batch> <out.npz>   |   check_evdq.py cmp a.npz b.npz"""
import os, sys, time
import numpy as np

def run(evdq, n, batch, out):
    os.environ["ASVD_EVDQ"] = str(evdq)
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from asvd4llm_amd import ops
    g = torch.Generator().manual_seed(7)
    mats = [(torch.randn(n, n, generator=g) / n ** 0.5).cuda() for _ in range(batch)]
    scales = [torch.rand(n, generator=g).add_(0.5).cuda() for _ in range(batch)]
    U, S, V, infos = ops.svd_batched(mats, scales)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        U, S, V, infos = ops.svd_batched(mats, scales)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    np.savez(out, S=torch.stack(S).cpu().numpy(), U=torch.stack(U).cpu().numpy(), V=torch.stack(V).cpu().numpy(), sweeps=np.array([i.sweeps for i in infos]))
    print(f"evdq={evdq} n={n} batch={batch} ms/call {1e3 * sorted(ts)[1]:.2f} sweeps {[i.sweeps for i in infos]} status {[i.status for i in infos]}", flush=True)

if sys.argv[1] == "run":
    run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = all(np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]) for k in ("S", "U", "V", "sweeps"))
    print("bit-identical:", same, "max |dS|/S0", float(np.abs(a["S"] - b["S"]).max() / a["S"].max()))
    sys.exit(0 if same else 1)
