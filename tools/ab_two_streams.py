"""A/B: one asvd_svd_batched call of B problems against G concurrent calls of B/G problems (one host thread + one stream each).

VERDICT r4 item 1: the eigen-solve launches (VALU-bound, no HBM) and the fused update + Gram launches (HBM path) alternate on ONE stream;
does the hardware overlap them when the batch is split over streams?  Also: the same call on a CU-MASKED stream
(hipExtStreamCreateWithCUMask) to see how the per-class times scale with the number of CUs a call may use (space partitioning).

  python tools/ab_two_streams.py [--n 4096] [--batch 32] [--groups 1,2,4] [--reps 3] [--masks 256,224,192,160,128]
Prints one JSON object per configuration."""
import argparse, ctypes, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asvd4llm_amd import ops, _lib


def problems(dev, B, m, n, seed=233):
    out = []
    for b in range(B):
        g = torch.Generator(device=dev).manual_seed(seed + b)
        W = torch.randn(m, n, generator=g, device=dev) * 0.02
        W[:, torch.randperm(n, generator=g, device=dev)[: max(1, n // 200)]] *= 20
        out.append(W)
    return out


def run_groups(mats, G, reps, streams=None, sizes=None, cus=None):
    """G host threads, each factorises its slice of the batch `reps` times on its own stream; returns wall seconds per rep (all groups).
    sizes: problems per group (default: equal); cus: CUs of every group's masked stream (the library sizes its launches for them)"""
    dev = mats[0].device
    B = len(mats)
    sizes = sizes or [B // G] * G
    offs = [sum(sizes[:i]) for i in range(G)]
    res = [None] * G
    errs = []
    bar = threading.Barrier(G + 1)

    def worker(i):
        try:
            st = streams[i] if streams else torch.cuda.Stream(device=dev)
            if cus:
                _lib.load(True).asvd_svd_set_call_cus(int(cus[i]))   # per host thread
            with torch.cuda.stream(st):
                bar.wait()
                for _ in range(reps):
                    res[i] = ops.svd_batched(mats[offs[i]:offs[i] + sizes[i]])
                st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            try:
                bar.abort()
            except Exception:  # noqa: BLE001
                pass

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(G)]
    for t in ts:
        t.start()
    torch.cuda.synchronize()
    bar.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if errs:
        raise RuntimeError(errs[0])
    return dt, res


def masked_stream(dev, bits):
    """stream restricted to the CUs whose bit is set in `bits` (list of 0/1, one per CU)"""
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        raise RuntimeError("libamdhip64 not found")
    words = (len(bits) + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i, b in enumerate(bits):
        if b:
            arr[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--groups", default="1,2,4")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--masks", default="")
    ap.add_argument("--mask_mode", default="first", help="first: the first N bits; stride: bits spread evenly over the 256")
    ap.add_argument("--warm_s", type=float, default=4.0)
    ap.add_argument("--mask_overlap", type=int, default=0)
    ap.add_argument("--parts", default="", help='";"-separated partitions, each "problems:CUs,problems:CUs,..." (masked streams, CU-aware plans)')
    ap.add_argument("--group_masks", action="store_true", help="give every group its own 1/G of the CUs (masked streams) instead of sharing the chip")
    args = ap.parse_args()
    _lib.load(True)
    dev = torch.device("cuda", 0)
    mats = problems(dev, args.batch, args.m, args.n)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.warm_s:
        ref = ops.svd_batched(mats)
        torch.cuda.synchronize()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for G in [int(x) for x in args.groups.split(",") if x]:
        # the SAME streams for the warm-up and the timed repetitions: the caching allocator keeps one pool per stream, a fresh stream would pay
        # hipMalloc for its 3 GB of outputs + workspace inside the timed region
        if args.group_masks:
            per_cu = ncu // G
            ov = args.mask_overlap   # every group additionally gets `ov` CUs of its neighbours (shared between two groups)
            streams = [masked_stream(dev, [1 if (i * per_cu - ov <= c < (i + 1) * per_cu + ov) else 0 for c in range(ncu)]) for i in range(G)]
        else:
            streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
        dt, res = run_groups(mats, G, 2, streams)           # allocator warm-up for this split
        dt, res = run_groups(mats, G, args.reps, streams)
        same = all(torch.equal(a, b) for i in range(G) for a, b in zip(res[i][1], ref[1][i * (args.batch // G):(i + 1) * (args.batch // G)]))
        print(json.dumps({"exp": "groups", "cu_masked_partition": bool(args.group_masks), "mask_overlap": args.mask_overlap, "m": args.m, "n": args.n, "batch": args.batch, "groups": G, "ms_per_batch": 1e3 * dt,
                          "svd_per_s": args.batch / dt, "sigma_bit_identical_to_one_call": bool(same),
                          "sweeps": sorted(set(i.sweeps for r in res for i in r[3]))}), flush=True)
    for spec in [x for x in args.parts.split(";") if x]:
        # "16:128,16:128" = two groups of 16 problems on 128 CUs each (consecutive CU ranges), the library told about the CU counts
        parts = [(int(a), int(b)) for a, b in (p.split(":") for p in spec.split(","))]
        sizes, cus = [p[0] for p in parts], [p[1] for p in parts]
        assert sum(sizes) == args.batch and sum(cus) <= ncu
        starts = [sum(cus[:i]) for i in range(len(cus))]
        streams = [masked_stream(dev, [1 if starts[i] <= c < starts[i] + cus[i] else 0 for c in range(ncu)]) for i in range(len(cus))]
        os.environ["ASVD_SPLIT"] = "0"
        dt, res = run_groups(mats, len(parts), 2, streams, sizes, cus)
        dt, res = run_groups(mats, len(parts), args.reps, streams, sizes, cus)
        print(json.dumps({"exp": "parts", "spec": spec, "m": args.m, "n": args.n, "batch": args.batch, "ms_per_batch": 1e3 * dt, "svd_per_s": args.batch / dt,
                          "sweeps": sorted(set(i.sweeps for r in res for i in r[3]))}), flush=True)
    for N in [int(x) for x in args.masks.split(",") if x]:
        if args.mask_mode == "first":
            bits = [1 if i < N else 0 for i in range(ncu)]
        else:
            bits = [0] * ncu
            for j in range(N):
                bits[(j * ncu) // N] = 1
        st = masked_stream(dev, bits)
        ops.svd_profile(True)
        with torch.cuda.stream(st):
            ops.svd_batched(mats)
            st.synchronize()
            t1 = time.perf_counter()
            ops.svd_batched(mats)
            st.synchronize()
            dt = time.perf_counter() - t1
        prof = ops.svd_profile()
        ops.svd_profile(False)
        cls = {k: {"ms": v["ms"], "launches": v["launches"], "avg_us": 1e3 * v["ms"] / max(1, v["launches"])} for k, v in prof.items()
               if isinstance(v, dict) and "ms" in v}
        print(json.dumps({"exp": "cu_mask", "mode": args.mask_mode, "cus": N, "of": ncu, "m": args.m, "n": args.n, "batch": args.batch,
                          "ms_per_batch": 1e3 * dt, "classes": cls}), flush=True)


if __name__ == "__main__":
    main()
