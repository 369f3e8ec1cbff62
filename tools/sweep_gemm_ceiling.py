"""What the PyTorch forwards of the sensitivity sweep can reach on this GPU, shape by shape (VERDICT r5 task 6: "several calibration samples per
suffix pass when HBM allows").  The sweep's suffix pass runs one decoder block after the other on [R x k, T, C] tokens — R = 6 candidate ratios,
k = calibration samples per pass (1 today), T = 2047 — through the reference's own modules (fp16 nn.Linear = hipBLASLt, SDPA, RMSNorm, SiLU).  This
tool times exactly those operators at M = R k T rows for k = 1, 2, 4 and prints TFLOP/s per operator and per block: if a block is not faster per
token at k = 2 / 4 than at k = 1, batching more samples per pass cannot shorten the sweep, whatever the evaluator does.

  python tools/sweep_gemm_ceiling.py > profiles/r6_sweep_gemm_ceiling.txt"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=10):
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    import torch
    import torch.nn.functional as Fn
    dev = torch.device("cuda", 0)
    C, I, H, T, V = 4096, 11008, 32, 2047, 32000
    g = torch.Generator(device=dev).manual_seed(1)
    Wq = (torch.randn(C, C, generator=g, device=dev) * 0.02).half()
    Wu = (torch.randn(I, C, generator=g, device=dev) * 0.02).half()
    Wd = (torch.randn(C, I, generator=g, device=dev) * 0.02).half()
    Wh = (torch.randn(V, C, generator=g, device=dev) * 0.02).half()
    print("# Llama-2-7B block operators at M = 6 k 2047 rows (fp16, torch on this GPU); seconds per call, TFLOP/s", flush=True)
    for k in (1, 2, 4):
        Bt = 6 * k
        x = (torch.randn(Bt, T, C, generator=g, device=dev)).half()
        xi = (torch.randn(Bt, T, I, generator=g, device=dev)).half()
        q = torch.randn(Bt, H, T, C // H, generator=g, device=dev).half()
        rec = {"samples_per_pass": k, "rows": Bt * T}
        t = timed(lambda: Fn.linear(x, Wq))
        rec["linear_4096x4096"] = {"s": t, "TFLOPs": 2.0 * Bt * T * C * C / t / 1e12}
        t2 = timed(lambda: Fn.linear(x, Wu))
        rec["linear_11008x4096"] = {"s": t2, "TFLOPs": 2.0 * Bt * T * C * I / t2 / 1e12}
        t3 = timed(lambda: Fn.linear(xi, Wd))
        rec["linear_4096x11008"] = {"s": t3, "TFLOPs": 2.0 * Bt * T * C * I / t3 / 1e12}
        t4 = timed(lambda: Fn.scaled_dot_product_attention(q, q, q, is_causal=True))
        rec["sdpa_causal"] = {"s": t4, "TFLOPs": 4.0 * Bt * H * T * T * (C // H) / 2 / t4 / 1e12}
        t5 = timed(lambda: Fn.silu(xi) * xi)
        rec["silu_mul_11008"] = {"s": t5, "GBps": 3.0 * xi.numel() * 2 / t5 / 1e9}
        t6 = timed(lambda: Fn.rms_norm(x, (C,)))
        rec["rms_norm"] = {"s": t6, "GBps": 2.0 * x.numel() * 2 / t6 / 1e9}
        block_s = 4 * t + 2 * t2 + t3 + t4 + t5 + 2 * t6
        block_flop = 2.0 * Bt * T * (4 * C * C + 3 * C * I) + 4.0 * Bt * H * T * T * (C // H) / 2
        rec["block_sum"] = {"s": block_s, "TFLOPs": block_flop / block_s / 1e12, "us_per_token": 1e6 * block_s / (Bt * T)}
        if k == 1:
            t7 = timed(lambda: Fn.linear(x, Wh), reps=3)
            rec["lm_head_32000x4096"] = {"s": t7, "TFLOPs": 2.0 * Bt * T * C * V / t7 / 1e12}
        print(json.dumps(rec), flush=True)
        del x, xi, q
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
