"""Turn the per-kernel PMC text summaries of tools/prof_final.sh into profiles/pmc_traffic.json (HBM bytes per launch of the streaming
kernels).  FETCH_SIZE is reported in KiB of 64-B requests: on gfx950 a wide coalesced streaming read is counted at HALF its bytes
(MI355X_MICROARCH.md, HBM section) -> doubled here; WRITE_SIZE is taken as reported (KiB)."""
import json, os, sys
d = sys.argv[1]
def table(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        p = line.split()
        if len(p) >= 5 and p[-3].isdigit():
            out[(p[0], p[1])] = {"calls": int(p[-3]), "avg": float(p[-2]), "avg_dur_us": float(p[-1])}
    return out
fetch, write = table(os.path.join(d, "pmc_FETCH_SIZE.txt")), table(os.path.join(d, "pmc_WRITE_SIZE.txt"))
names = {"supgram": "supgram_kernel", "supdate": "supdate_split_kernel", "sgram": "sgram6_kernel", "evd": "evdw12_kernel", "snapshot": "fullcheck_kernel"}
import hashlib
_lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "asvd4llm_amd", "libasvd_hip.so")
res = {"batch": int(os.environ.get("PMC_BATCH", "32")), "lib_sha256": hashlib.sha256(open(_lib_path, "rb").read()).hexdigest() if os.path.exists(_lib_path) else None, "source": "profiles/" + os.environ.get("PMC_TAG", os.path.basename(d).replace("prof_", "")) + "_pmc_*.txt",
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --no_cpu_baseline --no_latency --steps 1 --warmup 0 --prewarm_s 0` "
               "(" + os.environ.get("PMC_BATCH", "32") + " x 4096x4096 fp32, one stream); mean per dispatch over all launches of the run; FETCH doubled (gfx950 half-count), WRITE as reported",
       "kernels": {}}
import glob
def first(pattern):
    m = sorted(glob.glob(os.path.join(d, pattern)))
    return m[0] if m else os.path.join(d, "missing")
busy = table(first("pmc_SQ_VALU_MFMA_BUSY*.txt"))
grbm = table(first("pmc_GRBM_GUI_ACTIVE*.txt"))
for key, kn in names.items():
    f = [v for (n, c), v in fetch.items() if kn in n and c == "FETCH_SIZE"]
    w = [v for (n, c), v in write.items() if kn in n and c == "WRITE_SIZE"]
    if not f and not w:
        continue
    fk = sum(x["avg"] * x["calls"] for x in f) / max(1, sum(x["calls"] for x in f)) if f else 0.0
    wk = sum(x["avg"] * x["calls"] for x in w) / max(1, sum(x["calls"] for x in w)) if w else 0.0
    res["kernels"][key] = {"fetch_kib_raw": fk, "write_kib": wk, "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
                           "launches": sum(x["calls"] for x in f) if f else sum(x["calls"] for x in w)}
    # matrix-pipe occupancy and shader clock of the same kernel (their own passes): SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs of a shader
    # engine and averaged over the 32 engines by the summary (32 x 32 = 1024 SIMDs); GRBM_GUI_ACTIVE = shader cycles the dispatch was active
    mb = [v for (n, c), v in busy.items() if kn in n and c == "SQ_VALU_MFMA_BUSY_CYCLES"]
    ga = [v for (n, c), v in grbm.items() if kn in n and c == "GRBM_GUI_ACTIVE"]
    if mb and ga:
        mbusy = sum(x["avg"] * x["calls"] for x in mb) / max(1, sum(x["calls"] for x in mb))
        gact = sum(x["avg"] * x["calls"] for x in ga) / max(1, sum(x["calls"] for x in ga))
        gdur = sum(x["avg_dur_us"] * x["calls"] for x in ga) / max(1, sum(x["calls"] for x in ga))
        res["kernels"][key].update({"mfma_busy_cycles_avg": mbusy, "gui_active_cycles": gact, "mfma_busy_frac": (mbusy * 32 / 1024) / gact if gact else None,
                                    "shader_clock_GHz_under_pmc": gact / gdur / 1e3 if gdur else None})
# HBM bytes of EVERY kernel of the pass (2 x FETCH + WRITE, KiB -> B): the pass runs `bench.py --steps 1 --warmup 0 --prewarm_s 0`, i.e. PMC_STEPS
# identical steps of PMC_BATCH problems (the timed one + the profiled ones)
steps = int(os.environ.get("PMC_STEPS", "3"))
tot_f = sum(v["avg"] * v["calls"] for (n, c), v in fetch.items() if c == "FETCH_SIZE")
tot_w = sum(v["avg"] * v["calls"] for (n, c), v in write.items() if c == "WRITE_SIZE")
if tot_f and tot_w:
    res["hbm_bytes_all_kernels_of_the_pass"] = int((2 * tot_f + tot_w) * 1024)
    res["steps_in_the_pass"] = steps
    res["hbm_bytes_per_svd_all_kernels"] = (2 * tot_f + tot_w) * 1024 / steps / res["batch"]
print(json.dumps(res, indent=1))
