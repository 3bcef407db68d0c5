#!/usr/bin/env python3
"""Do kernels of a rocprofv3 (rocpd sqlite) kernel trace overlap in time?  For the last `n` dispatches: the union of the
busy intervals against the sum of the durations (equal = serial; sum > union = concurrent kernels), per queue counts.
usage: prof_overlap.py <results.db> [n]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
qcol = next((q for q in ("queue_id", "queue", "stream_id", "stream") if q in cols), None)
rows = c.execute(f"select start, end{', ' + qcol if qcol else ''} from kernels order by start desc limit {n}").fetchall()[::-1]
tot = sum(r[1] - r[0] for r in rows)
union, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for r in rows[1:]:
    if r[0] > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
union += cur_e - cur_s
span = max(r[1] for r in rows) - rows[0][0]
print(f"last {len(rows)} dispatches: span {span/1e6:.2f} ms, union of busy intervals {union/1e6:.2f} ms, sum of durations {tot/1e6:.2f} ms "
      f"-> {tot/union:.3f} kernels in flight on average while busy")
if qcol:
    qs = {}
    for r in rows:
        qs[r[2]] = qs.get(r[2], 0) + 1
    print(f"dispatches per {qcol}:", qs)
