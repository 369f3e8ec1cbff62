"""GB/s of the HBM-bound kernels (K1/K2 hook statistics, K3 scale, K5 truncate/split, K8 Frobenius) at Llama-2-7B shapes.
Algorithmic bytes per SURVEY.md 8(d); timing with HIP events on torch's current stream (the stream the kernels are launched on)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asvd4llm_amd import ops

dev = torch.device("cuda", 0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


out = []
for (T, C) in ((2048, 4096), (2048, 11008)):
    x = torch.randn(T, C, device=dev).half()
    acc = torch.zeros(C, device=dev).half()
    for method in ("abs_mean", "abs_max"):
        t = timeit(lambda: ops.absstat_accum(x, acc, method))
        b = T * C * 2 + 2 * C * 2
        out.append({"kernel": f"absstat_{method}", "shape": [T, C], "us": t * 1e6, "alg_bytes": b, "GBps": b / t / 1e9})
    # the reference's eager chain for the same hook update (3 kernels + a [T,C] temporary), torch on the same GPU
    def ref():
        a = x.abs().mean(dim=-2).view(-1)
        acc.add_(a)
    t = timeit(ref)
    out.append({"kernel": "torch_eager_abs_mean_chain (reference hook body on the same GPU)", "shape": [T, C], "us": t * 1e6, "alg_bytes": b, "GBps": b / t / 1e9})
for (m, n, r) in ((4096, 4096, 512), (4096, 4096, 1843), (11008, 4096, 2686)):
    k = min(m, n)
    U = torch.randn(m, k, device=dev); V = torch.randn(n, k, device=dev); S = torch.rand(k, device=dev).sort(descending=True).values
    s = (torch.rand(n, device=dev) + 0.5).half()
    t = timeit(lambda: ops.truncate_split(U, S, V, s, r, "UV", torch.float16))
    b = 4 * r * (m + n) + 4 * r + 2 * n + 2 * r * (m + n)
    out.append({"kernel": "truncate_split", "shape": [m, n, r], "us": t * 1e6, "alg_bytes": b, "GBps": b / t / 1e9})
    W = torch.randn(m, n, device=dev).half()
    t = timeit(lambda: ops.scale_cols(W, s))
    b = m * n * (2 + 4) + n * 2
    out.append({"kernel": "scale_cols", "shape": [m, n], "us": t * 1e6, "alg_bytes": b, "GBps": b / t / 1e9})
    t = timeit(lambda: ops.fro_norm_sq(W))
    out.append({"kernel": "fro_norm_sq", "shape": [m, n], "us": t * 1e6, "alg_bytes": m * n * 2, "GBps": m * n * 2 / t / 1e9})
# K10 one-launch SVDLinear forward against the reference's two nn.Linear launches (hipBLASLt), Llama-2-7B projection at ratio 0.9.
# Wall time per call on the stream (back-to-back calls, so launch cost is included for both); kernel-only times: rocprofv3 of this script.
import torch.nn as nn
for (K, r, N) in ((4096, 1843, 4096), (4096, 2686, 11008)):
    Bw = (torch.randn(r, K, device=dev) / K ** 0.5).half()
    Aw = (torch.randn(N, r, device=dev) / r ** 0.5).half()
    Ap, Bp, work = ops.lowrank_pack(Aw, Bw)
    for T in (1, 2, 4, 16, 64):
        x = torch.randn(T, K, device=dev).half()
        t_f = timeit(lambda: ops.lowrank_forward(x, Ap, Bp, None, work), reps=50)
        t_2 = timeit(lambda: nn.functional.linear(nn.functional.linear(x, Bw), Aw), reps=50)
        b = 2 * r * (K + N) + 2 * T * (K + N)
        out.append({"kernel": "lowrank_forward (K10)", "shape": [T, K, r, N], "us": t_f * 1e6, "us_two_nn_linear": t_2 * 1e6, "alg_bytes": b,
                    "GBps": b / t_f / 1e9, "GBps_two_nn_linear": b / t_2 / 1e9})
# K4s sigma_max (Lanczos): algorithmic bytes = 2 passes over W per step (the second pass re-reads the 64-row chunk from L2)
import time
for (m, n, B) in ((4096, 4096, 16), (11008, 4096, 16), (4096, 11008, 16), (4096, 4096, 1)):
    mats = [(torch.randn(m, n, device=dev) * 0.02).half() for _ in range(B)]
    ops.sigma_max_batched(mats)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sig, info = ops.sigma_max_batched(mats)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    steps = max(i[1] for i in info)
    b = 2.0 * m * n * 2 * B * steps
    _, S, _, _ = ops.svd_batched(mats[:1], None, k=1, want_vectors=False)
    out.append({"kernel": "sigma_max_lanczos", "shape": [m, n], "batch": B, "lanczos_steps": steps, "ms_total": t * 1e3, "ms_per_matrix": t * 1e3 / B,
                "alg_bytes": b, "GBps": b / t / 1e9, "rel_diff_vs_jacobi_k1": abs(sig[0].item() - S[0][0].item()) / S[0][0].item()})
for o in out:
    print(json.dumps(o))
