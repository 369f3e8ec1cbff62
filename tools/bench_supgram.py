"""Timing of ONE launch shape of the fused update + Gram kernel (supgram_kernel) in isolation.

  python tools/bench_supgram.py [--near_identity] [--timing]

--near_identity  Q = qr(I + 1e-3 N): the late-sweep regime (most of a real step); default is a random orthogonal Q (all three bf16
                 parts of Q dense: the worst case for the matrix pipe's power draw)
--timing         measurement build (-DASVD_SG_TIMING -> asvd4llm_amd/libasvd_hip_meas.so, built here beforehand with
                 `python tools/bench_supgram.py --build_meas`): s_memtime stamps of one wave per pair of workgroup (0,0,0) —
                 compute segment (Gram part | update part), barrier wait, memory segment, barrier wait, in shader cycles per tile."""
import argparse, ctypes, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MEAS = os.path.join(ROOT, "asvd4llm_amd", "libasvd_hip_meas.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--near_identity", action="store_true")
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--build_meas", action="store_true")
    ap.add_argument("--ns", type=int, default=64)
    ap.add_argument("--R", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--warm_s", type=float, default=1.5)
    ap.add_argument("--pad_rows", type=int, default=0, help="extra rows per panel in the allocation: panel stride (R + pad_rows) * 128 B instead of a power of two")
    ap.add_argument("--ablate", type=int, default=0, help="measurement build only (timing-only, results wrong): 1 no panel stores, 2 no fetch, 4 no matrix instructions, 8 no LDS operand stores")
    a = ap.parse_args()
    if a.build_meas:
        from asvd4llm_amd import build as b
        print(b.build(force=True, verbose=False, extra_flags=["-DASVD_SG_TIMING"], out=MEAS))
        return
    import torch
    from asvd4llm_amd import _lib as L
    if a.ablate:
        os.environ["ASVD_SG_ABLATE"] = str(a.ablate)
    if a.timing:
        lib = ctypes.CDLL(MEAS)
        lib.asvd_test_supgram.restype = ctypes.c_int
        lib.asvd_test_supgram.argtypes = L.SIGNATURES["asvd_test_supgram"][1]
    else:
        lib = L.load(True)
    gpu = torch.device("cuda:0")
    ns, R, batch = a.ns, a.R, a.batch
    nb, npairs = 2 * ns, ns // 2
    Rs = R + a.pad_rows   # rows allocated per panel
    Xfull = torch.randn(batch, nb, Rs, 32, device=gpu) * 0.05
    X = Xfull[:, :, :R]
    if a.near_identity:
        Q = torch.linalg.qr(torch.eye(128, device=gpu) + 1e-3 * torch.randn(npairs, 128, 128, device=gpu))[0]
    else:
        Q = torch.linalg.qr(torch.randn(npairs, 128, 128, device=gpu))[0]
    Q = Q.contiguous().unsqueeze(0).expand(batch, -1, -1, -1).contiguous()
    flags = torch.ones(batch, npairs, 4, dtype=torch.int32, device=gpu)
    done = torch.zeros(batch, dtype=torch.int32, device=gpu)
    nupd = torch.zeros(batch, dtype=torch.int32, device=gpu)
    Gx = torch.zeros(batch, npairs, 1, 6, 1024, device=gpu)
    nrm2 = (X.double() ** 2).sum(dim=2)   # squared column norms, [batch, nb, 32]; step D = 1 pairs super-panels (2k, 2k + 1): columns 128 k .. + 127
    Din = nrm2.reshape(batch, npairs, 128).float().contiguous()
    vp = ctypes.c_void_p
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.asvd_test_supgram(vp(Xfull.data_ptr()), Rs * 32, nb * Rs * 32, ns, 1, 2, R, R, R, vp(Q.data_ptr()), vp(flags.data_ptr()),
                                   vp(Din.data_ptr()), vp(Gx.data_ptr()), vp(done.data_ptr()), vp(nupd.data_ptr()), 1, npairs, batch, vp(st))
        assert rc == 0
    import time
    t_w = time.time()
    while time.time() - t_w < a.warm_s:   # DVFS settles over hundreds of milliseconds: a 10-launch measurement from idle reads the clock ramp, not the kernel
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    gb = 2 * X.numel() * 4 / 1e9
    tiles = (ns // 4) * batch * (R // 32) / 256.0   # 32-row tiles per CU
    out = {"near_identity": a.near_identity, "pad_rows": a.pad_rows, "lib": os.path.basename(os.environ.get("ASVD_HIP_LIB", "default")), "ablate": a.ablate, "us_per_launch": round(us, 1), "us_per_tile": round(us / tiles, 3), "TBps_rw": round(gb / us * 1e3, 3)}
    if a.timing:
        buf = (ctypes.c_ulonglong * 20)()
        lib.asvd_test_sg_timing.restype = ctypes.c_int
        assert lib.asvd_test_sg_timing(buf) == 0
        names = ["loop", "gram", "update", "barrier_1", "stash", "fetch", "-", "opnd_split+stores", "barrier_2"]
        for pr in range(2):
            n = max(1, int(buf[pr * 10 + 9]))
            out[f"pair{pr}_cycles_per_tile"] = {names[i]: round(buf[pr * 10 + i] / n) for i in range(9)}
            out[f"pair{pr}_cycles_per_tile"]["total"] = round(sum(buf[pr * 10 + i] for i in range(9)) / n)
        out["shader_clock_GHz"] = round(out["pair0_cycles_per_tile"]["total"] / (us / tiles) / 1e3, 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
