"""Summarise a rocprofv3 rocpd sqlite db: per-kernel count / total / avg.  usage: prof_summary.py db [skip_first_n_per_kernel]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:10.3f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:9.1f} {100*r[2]/tot:6.2f}")
print(f"total kernel time {tot/1e3:.3f} ms")
