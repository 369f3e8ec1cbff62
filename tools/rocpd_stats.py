"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel stats table (name, calls, total/avg/min/max us, %)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
q = f"""select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"""
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for n, c, t, a, mn, mx in rows:
    n = n.split("(")[0][-70:]
    print(f"{n:70s} {c:8d} {t/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*t/tot:6.2f}")
