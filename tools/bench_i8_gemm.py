"""Micro-benchmark of the int8 matrix-pipe kernels of round 6 (csrc/gram_i8.h, csrc/nn_gemm_i8.h) at the bench shape, 32 x 4096^2:
the Gram matrix of the reduction alone through the test hook (fp64 matrix instructions / int8 digit path, block orders 0 / 1 / 2), and the
reduction ("pack") and finalize classes of whole profiled asvd_svd_batched calls with the long-side product on either pipe.
One JSON object per line.  Knobs: ASVD_GI_ORDER / ASVD_NI_ORDER (measurement only), ASVD_GRAM_I8, ASVD_NN_I8."""
import ctypes, json, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asvd4llm_amd import _lib as L, ops  # noqa: E402


def time_gram(lib, P, G, scratch, ex, nb, m_pad, batch, mode, reps=4):
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = lib.asvd_test_gram(ctypes.c_void_p(P.data_ptr()), m_pad * 32, nb * m_pad * 32, nb, m_pad, batch, mode, 0, ctypes.c_void_p(G.data_ptr()),
                                ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.c_void_p(ex.data_ptr()), ctypes.c_void_p(st))
        assert rc == 0
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lib = L.load(True)
    dev = torch.device("cuda", 0)
    batch, n = int(os.environ.get("B", 32)), int(os.environ.get("N", 4096))
    m = int(os.environ.get("M", n))
    nb, m_pad = n // 32, (m + 31) // 32 * 32
    g = torch.Generator(device=dev).manual_seed(0)
    P = torch.randn(batch, nb, m_pad, 32, device=dev, generator=g) * 0.02
    G = torch.empty(batch, n, n, dtype=torch.float64, device=dev)
    scratch = torch.empty(3 * n * ((m_pad + 63) // 64 * 64) * batch, dtype=torch.int8, device=dev)
    ex = torch.zeros(batch, n, dtype=torch.int32, device=dev)
    ops_per = 9 * 2 * (n * n * m / 2) * batch
    out = {"what": "gram", "batch": batch, "m": m, "n": n, "fp64_ms": time_gram(lib, P, G, scratch, ex, nb, m_pad, batch, 0)}
    for order in (0, 1, 2):
        os.environ["ASVD_GI_ORDER"] = str(order)
        t = time_gram(lib, P, G, scratch, ex, nb, m_pad, batch, 1)
        out[f"i8_order{order}_ms_incl_colmax_split"] = t
        out[f"i8_order{order}_Pops_if_all_gemm"] = ops_per / t / 1e12
    os.environ.pop("ASVD_GI_ORDER")
    print(json.dumps(out), flush=True)
    del P, G, scratch
    torch.cuda.empty_cache()
    if os.environ.get("SKIP_SVD"):
        return
    mats = [torch.randn(m, n, device=dev, generator=g) * 0.02 for _ in range(batch)]
    scales = [1 + 30 * torch.rand(n, device=dev, generator=g) ** 8 for _ in range(batch)]
    os.environ["ASVD_SPLIT"] = "0"
    for name, env in (("bf16_long_side", {"ASVD_NN_I8": "0"}), ("i8_order1", {"ASVD_NI_ORDER": "1"}), ("i8_order0", {"ASVD_NI_ORDER": "0"}),
                      ("i8_order2", {"ASVD_NI_ORDER": "2"}), ("gram_fp64", {"ASVD_GRAM_I8": "0"}), ("gram_i8_order1", {"ASVD_GI_ORDER": "1"})):
        for k, v in env.items():
            os.environ[k] = v
        ops.svd_batched(mats, scales)
        ops.svd_profile(True)
        ops.svd_batched(mats, scales)
        pr = ops.svd_profile()
        ops.svd_profile(False)
        for k in env:
            os.environ.pop(k)
        print(json.dumps({"what": "svd_classes_unsplit", "variant": name, "pack_ms": pr["pack"]["ms"], "finalize_ms": pr["finalize"]["ms"],
                          "snapshot_ms": pr["snapshot"]["ms"], "total_ms": sum(v["ms"] for k, v in pr.items() if isinstance(v, dict) and "ms" in v)}), flush=True)


if __name__ == "__main__":
    main()
