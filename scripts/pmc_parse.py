#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE (KiB) over the GEMM dispatches of a rocprofv3 --pmc run (rocpd sqlite) and
apply the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced reads).
usage: pmc_parse.py results.db n_forwards"""
import json, sqlite3, sys
db, nfwd = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]
view = "counters_collection" if "counters_collection" in tabs else None
out = {"tables": pmc[:12]}
if view:
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    out["cols"] = cols
    kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "kernel" in x.lower()][0]
    ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x.lower() and "name" in x.lower()][0]
    vcol = "value" if "value" in cols else [x for x in cols if "value" in x.lower()][0]
    agg = {}
    for k, cn, v in c.execute(f"select {kcol}, {ccol}, {vcol} from {view}"):
        if "gemm" not in k:
            continue
        a = agg.setdefault(cn, [0.0, 0])
        a[0] += float(v); a[1] += 1
    out["sums"] = {k: {"sum": v[0], "dispatch_rows": v[1]} for k, v in agg.items()}
    if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
        fetch_b = agg["FETCH_SIZE"][0] * 1024.0 * 2.0          # KiB -> B, x2 gfx950 correction
        write_b = agg["WRITE_SIZE"][0] * 1024.0
        out["hbm_bytes_per_unet_fwd_b2"] = (fetch_b + write_b) / nfwd
print(json.dumps(out, indent=1))
