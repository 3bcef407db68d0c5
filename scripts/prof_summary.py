#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average, and
per-(kernel, grid) breakdown.  usage: prof_summary.py <results.db> [top_n]"""
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
s, e = c.execute("select min(start), max(end) from kernels").fetchone()
print(f"# kernels: {sum(r[1] for r in rows)} dispatches, busy {tot/1e6:.1f} ms, first-to-last span {(e-s)/1e6:.1f} ms")
print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for r in rows[:top]:
    print(f"{r[0][:80]:80s} {r[1]:7d} {r[2]/1e6:10.2f} {r[3]/1e3:9.1f} {100*r[2]/tot:6.1f}")
print("\n# per (kernel, grid) -- top 40 by total time")
rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), sum(duration), avg(duration) from kernels "
                 "group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 40").fetchall()
for r in rows:
    print(f"{r[0][:60]:60s} grid({r[1]},{r[2]},{r[3]}) calls {r[4]:6d} total {r[5]/1e6:9.2f} ms avg {r[6]/1e3:9.1f} us")
