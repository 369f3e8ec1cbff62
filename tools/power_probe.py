"""Socket power / shader clock / power cap telemetry next to a workload ("put telemetry in the evidence").

  python tools/power_probe.py --workload {idle,supgram,supgram_ni,bench,mfma,gram_i8,gram_fp64} [--seconds 6] [--out file.json]

A sampler thread reads amdsmi (power, gfx clock, temperature, power cap) every ~50 ms while the main thread keeps the GPU busy with
the chosen workload: back-to-back launches of the fused update + Gram kernel on random / near-identity Q (tools/bench_supgram.py's
shape), whole bench steps (32 x 4096^2 SVDs), or a dense bf16 GEMM loop (torch.matmul, context only: what the chip sustains on a plain
matrix-pipe load).  Prints one JSON line: mean / max power, mean clock, cap, samples, work done, joules per unit of work."""
import argparse, ctypes, json, os, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler(threading.Thread):
    def __init__(self, period=0.05):
        super().__init__(daemon=True)
        import amdsmi
        self.a = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]
        self.period, self.stop_flag, self.rows = period, False, []
        self.cap = None
        try:
            c = amdsmi.amdsmi_get_power_cap_info(self.h)
            self.cap = {k: c[k] for k in c}
        except Exception as e:  # noqa
            self.cap = {"error": str(e)}

    def sample(self):
        a, h = self.a, self.h
        row = {"t": time.time()}
        try:
            p = a.amdsmi_get_power_info(h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                if k in p and isinstance(p[k], (int, float)):
                    row["power_w"] = float(p[k]); break
        except Exception as e:  # noqa
            row["power_err"] = str(e)
        try:
            c = a.amdsmi_get_clock_info(h, a.AmdSmiClkType.GFX)
            row["sclk_mhz"] = float(c.get("clk", c.get("cur_clk", 0)))
        except Exception as e:  # noqa
            row["clk_err"] = str(e)
        try:
            m = a.amdsmi_get_gpu_metrics_info(h)
            for k in ("current_gfxclk", "average_gfxclk_frequency", "current_socket_power", "average_socket_power", "temperature_hotspot", "throttle_status", "indep_throttle_status"):
                if k in m and isinstance(m[k], (int, float)):
                    row["m_" + k] = m[k]
            if "current_gfxclks" in m:
                v = [x for x in m["current_gfxclks"] if isinstance(x, (int, float)) and 0 < x < 60000]
                if v:
                    row["m_gfxclk_xcd_mean"] = sum(v) / len(v)
        except Exception as e:  # noqa
            row["metrics_err"] = str(e)
        return row

    def run(self):
        while not self.stop_flag:
            self.rows.append(self.sample())
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="supgram")
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from asvd4llm_amd import _lib as L
    gpu = torch.device("cuda:0")
    work = None
    unit = "launch"
    if a.workload.startswith("supgram"):
        lib = L.load(True)
        ns, R, batch = 64, 4096, 32
        nb, npairs = 2 * ns, ns // 2
        X = torch.randn(batch, nb, R, 32, device=gpu) * 0.05
        if a.workload == "supgram_ni":
            Q = torch.linalg.qr(torch.eye(128, device=gpu) + 1e-3 * torch.randn(npairs, 128, 128, device=gpu))[0]
        else:
            Q = torch.linalg.qr(torch.randn(npairs, 128, 128, device=gpu))[0]
        Q = Q.contiguous().unsqueeze(0).expand(batch, -1, -1, -1).contiguous()
        flags = torch.ones(batch, npairs, 4, dtype=torch.int32, device=gpu)
        done = torch.zeros(batch, dtype=torch.int32, device=gpu)
        nupd = torch.zeros(batch, dtype=torch.int32, device=gpu)
        Gx = torch.zeros(batch, npairs, 1, 6, 1024, device=gpu)
        Din = (X.double() ** 2).sum(dim=2).reshape(batch, npairs, 128).float().contiguous()   # step D = 1 pairs super-panels (2k, 2k + 1)
        vp = ctypes.c_void_p
        st = torch.cuda.current_stream().cuda_stream

        def work():
            for _ in range(20):
                lib.asvd_test_supgram(vp(X.data_ptr()), R * 32, nb * R * 32, ns, 1, 2, R, R, R, vp(Q.data_ptr()), vp(flags.data_ptr()),
                                      vp(Din.data_ptr()), vp(Gx.data_ptr()), vp(done.data_ptr()), vp(nupd.data_ptr()), 1, npairs, batch, vp(st))
            torch.cuda.synchronize()
            return 20
    elif a.workload == "mfma":
        A = torch.randn(8192, 8192, device=gpu, dtype=torch.bfloat16)
        B = torch.randn(8192, 8192, device=gpu, dtype=torch.bfloat16)
        unit = "8192^3 bf16 GEMM"

        def work():
            for _ in range(10):
                torch.matmul(A, B)
            torch.cuda.synchronize()
            return 10
    elif a.workload == "bench":
        import bench as B
        from asvd4llm_amd import ops
        unit = "step of 32 SVDs"
        mats, stats = [], []
        for b in range(32):
            W, scal = B.synth(4096, 4096, seed=233 + b)
            mats.append(W.to(gpu)); stats.append(scal.to(gpu))

        def work():
            scales = ops.make_scale_batched(stats, alpha=0.5)
            U, S, V, infos = ops.svd_batched(mats, scales)
            ops.truncate_split_batched(U, S, V, scales, 512, "UV", torch.float16)
            torch.cuda.synchronize()
            return 1
    elif a.workload in ("gram_i8", "gram_fp64"):
        # the Gram matrix of the reduction alone (asvd_test_gram): int8 digit path / fp64 matrix instructions, 32 x 4096^2, back to back
        from asvd4llm_amd import _lib as L
        lib = L.load(True)
        vp = ctypes.c_void_p
        gb, gn = 32, 4096   # (`n` is the loop counter of main)
        gnb = gn // 32
        P = torch.randn(gb, gnb, gn, 32, device=gpu) * 0.02
        G = torch.empty(gb, gn, gn, dtype=torch.float64, device=gpu)
        scratch = torch.empty(3 * gn * gn * gb, dtype=torch.int8, device=gpu)
        ex = torch.zeros(gb, gn, dtype=torch.int32, device=gpu)
        mode = 1 if a.workload == "gram_i8" else 0
        unit = "Gram matrix of 32 x 4096^2 (" + ("int8 digit planes incl. digitising" if mode else "fp64 MFMA") + ")"
        st = torch.cuda.current_stream().cuda_stream

        def work():
            for _ in range(10):
                lib.asvd_test_gram(vp(P.data_ptr()), gn * 32, gnb * gn * 32, gnb, gn, gb, mode, 0, vp(G.data_ptr()), vp(scratch.data_ptr()), scratch.numel(),
                                   vp(ex.data_ptr()), vp(st))
            torch.cuda.synchronize()
            return 10
    elif a.workload == "idle":
        def work():
            time.sleep(0.2)
            return 0
    if work is not None:
        work()  # warm
    s = Sampler()
    s.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        n += work()
    t1 = time.time()
    s.stop_flag = True
    s.join()
    rows = [r for r in s.rows if r["t"] >= t0 + 0.5]  # skip the ramp
    def mean(k):
        v = [r[k] for r in rows if k in r]
        return round(sum(v) / len(v), 1) if v else None
    def mx(k):
        v = [r[k] for r in rows if k in r]
        return max(v) if v else None
    pw = mean("power_w") or mean("m_current_socket_power") or mean("m_average_socket_power")
    out = {"workload": a.workload, "seconds": round(t1 - t0, 2), "units": n, "unit": unit, "samples": len(rows), "power_w_mean": pw, "power_w_max": mx("power_w") or mx("m_current_socket_power"),
           "sclk_mhz_mean": mean("sclk_mhz"), "metrics_gfxclk_mean": mean("m_current_gfxclk") or mean("m_average_gfxclk_frequency"), "metrics_gfxclk_xcd_mean": mean("m_gfxclk_xcd_mean"),
           "hotspot_c_max": mx("m_temperature_hotspot"), "throttle_status": mx("m_throttle_status"), "power_cap": s.cap,
           "us_per_unit": round((t1 - t0) / n * 1e6, 1) if n else None, "joules_per_unit": round(pw * (t1 - t0) / n, 4) if (n and pw) else None,
           "first_row": s.rows[0] if s.rows else None}
    line = json.dumps(out, default=str)
    print(line, flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
