"""How much do kernels overlap in time? (rocprofv3 rocpd .db)  Prints busy time (union of intervals), sum of durations, per-queue stats."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else None
rows = list(cur.execute(f"select start, end, {qcol or 0}, kernel_id from rocpd_kernel_dispatch order by start"))
tot = sum(e - s for s, e, *_ in rows)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, *_ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"dispatches={len(rows)} sum_durations={tot/1e6:.1f} ms union_busy={busy/1e6:.1f} ms span={span/1e6:.1f} ms overlap_factor={tot/busy:.3f}")
from collections import Counter
print("queues:", Counter(r[2] for r in rows))
