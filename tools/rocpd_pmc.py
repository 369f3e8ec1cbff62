"""Per-kernel PMC summary from a rocprofv3 rocpd .db (one counter per pass): mean counter value per dispatch and mean duration."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pc = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
ic = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
kc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kc else "display_name"
q = f"""select s.{name_col}, p.name, count(*), avg(e.value), sum(e.value), avg(d.end - d.start)
        from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.{name_col}, p.name order by 5 desc"""
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print("query failed:", ex, "\npmc_event cols", pc, "\ninfo_pmc cols", ic); sys.exit(1)
print(f"{'kernel':60s} {'counter':12s} {'calls':>7s} {'avg_value':>14s} {'avg_dur_us':>10s}")
for n, c, cnt, avg, tot, dur in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{n.split('(')[0][-60:]:60s} {c:12s} {cnt:7d} {avg:14.1f} {dur/1e3:10.1f}")
