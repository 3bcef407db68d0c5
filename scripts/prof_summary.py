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

# ---- per DDIM step (between consecutive ddim_apply_kernel launches = one guided UNet forward + the step): busy time,
# idle gaps between kernels, and the per-kernel shares averaged over the steps
marks = [r[0] for r in c.execute("select start from kernels where name like '%ddim_apply_kernel%' order by start").fetchall()]
if len(marks) >= 3:
    spans = []
    agg = {}
    for a, b in zip(marks[1:-1], marks[2:]):       # skip the first interval (graph capture / warm-up)
        ks = c.execute("select name, start, end from kernels where start >= ? and start < ? order by start", (a, b)).fetchall()
        busy = sum(k[2] - k[1] for k in ks)
        gaps = sum(max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1))
        spans.append((b - a, busy, gaps, len(ks)))
        for k in ks:
            e = agg.setdefault(k[0], [0, 0])
            e[0] += 1
            e[1] += k[2] - k[1]
    n = len(spans)
    print(f"\n# per DDIM step ({n} steps between ddim_apply_kernel launches): span {sum(s[0] for s in spans)/n/1e6:.2f} ms, "
          f"kernel-busy {sum(s[1] for s in spans)/n/1e6:.2f} ms, gaps between kernels {sum(s[2] for s in spans)/n/1e6:.2f} ms, "
          f"{sum(s[3] for s in spans)/n:.0f} kernels")
    for name, (cnt, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:80]:80s} {cnt/n:7.1f} {dur/n/1e6:10.3f} ms {dur/cnt/1e3:9.1f} us {100*dur/sum(s[1] for s in spans):6.1f}")
