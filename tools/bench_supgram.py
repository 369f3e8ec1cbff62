"""Timing of ONE launch shape of the fused update + Gram kernel (supgram_kernel) in isolation, with timing-only ablations
(ASVD_SG_ABLATE bits: 1 no panel stores, 2 one of eight update k-steps, 4 no Gram MFMAs, 8 no panel fetch, 16 no Gram operand split,
32 no next-tile split).  The ablated results are wrong by construction; the numbers only say which resource the kernel waits for."""
import ctypes, json, os, sys, subprocess

def one(abl, ns=64, R=4096, batch=32, D=1, E=2, reps=10):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from asvd4llm_amd import _lib as L
    os.environ["ASVD_SG_ABLATE"] = str(abl)
    lib = L.load(True)
    gpu = torch.device("cuda:0")
    nb, npairs = 2 * ns, ns // 2
    X = torch.randn(batch, nb, R, 32, device=gpu) * 0.05
    Q = torch.linalg.qr(torch.randn(npairs, 128, 128, device=gpu))[0].contiguous().unsqueeze(0).expand(batch, -1, -1, -1).contiguous()
    flags = torch.ones(batch, npairs, 4, dtype=torch.int32, device=gpu)
    done = torch.zeros(batch, dtype=torch.int32, device=gpu)
    nupd = torch.zeros(batch, dtype=torch.int32, device=gpu)
    Gx = torch.zeros(batch, npairs, 1, 6, 1024, device=gpu)
    vp = ctypes.c_void_p
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.asvd_test_supgram(vp(X.data_ptr()), R * 32, nb * R * 32, ns, D, E, R, R, R, vp(Q.data_ptr()), vp(flags.data_ptr()),
                                   vp(Gx.data_ptr()), vp(done.data_ptr()), vp(nupd.data_ptr()), 1, npairs, batch, vp(st))
        assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gb = 2 * X.numel() * 4 / 1e9
    print(json.dumps({"ablate": abl, "us_per_launch": round(us, 1), "us_per_tile": round(us / (2 * R / 32), 3), "TBps_rw": round(gb / us * 1e3 / 1e3, 3)}), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]))
    else:
        for abl in [0, 1, 2, 4, 8, 16, 32, 6, 7, 9, 15, 63, 54]:
            subprocess.run([sys.executable, os.path.abspath(__file__), str(abl)], check=False)
