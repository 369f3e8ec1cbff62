"""Per-stream timeline of the dense sweeps from a rocprofv3 rocpd .db: for one queue, the sequence of kernels of a few consecutive
super-steps with start offsets, durations and the idle gap before each (what the stream's dependency chain looks like)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.{name_col}, d.grid_size_x, d.grid_size_y, d.grid_size_z from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
def short(n):
    n = n.split("(")[0]
    for k in ("sgram", "supdate_split", "supdate", "gupdate", "sassemble", "sfinish", "evd", "gram_kernel", "update_kernel", "fullcheck", "solve", "smid", "send"):
        if k in n: return k
    return n[-24:]
from collections import Counter
qs = Counter(r[2] for r in rows)
q = [k for k, v in qs.most_common(4)][1]  # one of the group streams
sel = [r for r in rows if r[2] == q]
# find the first sgram of the second half (a dense sweep in the profiled step) and print ~40 kernels from there
idx = [i for i, r in enumerate(sel) if short(r[3]) == "sgram"]
if idx:
    i0 = idx[len(idx) // 2]
    t0 = sel[i0][0]
    prev_end = sel[i0 - 1][1] if i0 > 0 else t0
    for r in sel[i0:i0 + 40]:
        print(f"{(r[0]-t0)/1e3:9.1f} us  gap {max(0,(r[0]-prev_end))/1e3:7.1f}  dur {(r[1]-r[0])/1e3:8.1f}  {short(r[3]):14s} grid {r[4]}x{r[5]}x{r[6]}")
        prev_end = r[1]
# per-kernel average gap before it (same queue) and duration, dense part only
import collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for qq in qs:
    s2 = [r for r in rows if r[2] == qq]
    for a, b in zip(s2, s2[1:]):
        k = short(b[3]); agg[k][0] += 1; agg[k][1] += max(0, b[0] - a[1]) / 1e3; agg[k][2] += (b[1] - b[0]) / 1e3
print("kernel          calls  avg_gap_before_us  avg_dur_us")
for k, (n, g, d) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:14s} {n:6d} {g/n:12.1f} {d/n:12.1f}")
