#!/usr/bin/env python3
"""Per-kernel counter averages from a rocprofv3 --pmc run (rocpd sqlite): {kernel [grid]: {counter: mean per
dispatch}}.  usage: pmc_dump.py results.db [name-filter]"""
import json, sqlite3, sys
db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if not view:
    print(json.dumps({"error": "no counters_collection view", "tables": tabs[:40]})); sys.exit(0)
cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "kernel" in x.lower()][0]
ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x.lower() and "name" in x.lower()][0]
vcol = "value" if "value" in cols else [x for x in cols if "value" in x.lower()][0]
gcols = [x for x in ("grid_size", "grid_size_x", "grid_x") if x in cols]
dcol = [x for x in ("dispatch_id", "id") if x in cols]
sel = f"select {kcol}, {ccol}, {vcol}" + (f", {gcols[0]}" if gcols else ", 0") + (f", {dcol[0]}" if dcol else ", 0") + f" from {view}"
agg = {}
for k, cn, v, g, d in c.execute(sel):
    if flt and flt not in k:
        continue
    key = f"{k[:70]} grid={g}"
    a = agg.setdefault(key, {}).setdefault(cn, {})
    a[d] = a.get(d, 0.0) + float(v)              # one dispatch may report several rows (per XCD / SE): sum them
out = {k: {cn: sum(dd.values()) / max(len(dd), 1) for cn, dd in v.items()} for k, v in agg.items()}
print(json.dumps(out, indent=1))
