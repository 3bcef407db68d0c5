#!/usr/bin/env python3
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh per kernel family and write the two JSON files
bench.py quotes (profiles/r06_pmc_unet_traffic.json, r06_pmc_gn_traffic.json).
usage: pmc_traffic.py fetch.db write.db n_forwards out_dir"""
import json
import os
import sqlite3
import sys


def sums(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counter" in t.lower()][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "kernel" in x.lower()][0]
    ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x.lower() and "name" in x.lower()][0]
    vcol = "value" if "value" in cols else [x for x in cols if "value" in x.lower()][0]
    agg = {}
    for k, cn, v in c.execute(f"select {kcol}, {ccol}, {vcol} from {view}"):
        if cn != counter:
            continue
        fam = "gemm" if ("gemm" in k and "splitk" not in k) or "ff_fused" in k or "tb_fused" in k or "conv_halo" in k or "qkv_attn" in k else "gn" if ("gn_" in k) else None
        if fam is None:
            continue
        a = agg.setdefault(fam, {"sum": 0.0, "rows": 0, "kernels": {}})
        a["sum"] += float(v)
        a["rows"] += 1
        a["kernels"][k[:60]] = a["kernels"].get(k[:60], 0) + 1
    return agg


fetch_db, write_db, nfwd, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
f, w = sums(fetch_db, "FETCH_SIZE"), sums(write_db, "WRITE_SIZE")
CORR = ("MI355X_MICROARCH.md HBM section: counters in KiB; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B); WRITE_SIZE as "
        "reported; Infinity-Cache hits are counted, so this is L2-miss (fabric) traffic, an upper bound on HBM bytes")
# algorithmic bytes of the GEMM family per B=2 forward: 43.2 GB with every projection its own GEMM; the one-launch level-0
# feed-forward (tc_ff_geglu_fused) reads and writes its rows once -- 10 x (576.7 - 104.9) MB less; since round 6 every temporal
# self-attention's projection lives in tc_temporal_qkv_attn (counted with the family), which stores [rows, C] where the qkv GEMM
# stored [rows, 3C]: 10 x 104.9 (level 0) + 2 x 83.9 (init_attn, C = 512) + 10 x 52.4 (level 1) + 10 x 26.2 (level 2) + 2 x 6.6
# (middle block) = 2016 MB less (round 5 had the level-0 single launch instead: 10 x 262.0 MB less)
for fam, fname, algo, unit in (("gemm", "r06_pmc_unet_traffic.json", 43.2e9 - 10 * 471.9e6 - 2016.0e6,
                                "tc_gemm_bf16 / tc_ff_geglu_fused / tc_temporal_qkv_attn launch"),
                               ("gn", "r06_pmc_gn_traffic.json", 8.487e9, "tc_groupnorm call")):
    if fam not in f or fam not in w:
        print("no rows for", fam, file=sys.stderr)
        continue
    rd = f[fam]["sum"] * 1024.0 * 2.0 / nfwd
    wr = w[fam]["sum"] * 1024.0 / nfwd
    disp = f[fam]["rows"] / nfwd
    calls = 166.0 if fam == "gn" else disp          # a tc_groupnorm call is 1 or 3 dispatches: 166 calls per B=2 forward
    out = {"what": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/pmc_unet.py: {nfwd} eager "
                   f"B=2 UNet forwards of the bench model at the round-6 HEAD (ABI 13: temporal qkv projection + attention as one launch at every level); sums over the {fam} kernel dispatches",
           "correction": CORR, "forwards": nfwd, "dispatches_per_forward": disp, "kernels": f[fam]["kernels"],
           "fetch_size_kib_sum": f[fam]["sum"], "write_size_kib_sum": w[fam]["sum"],
           "fabric_read_bytes_per_forward": rd, "fabric_write_bytes_per_forward": wr,
           "traffic_bytes_per_forward": rd + wr, "traffic_bytes_per_launch": (rd + wr) / calls,
           "per": unit, "algorithmic_bytes_per_b2_forward": algo, "traffic_over_algorithmic": (rd + wr) / algo}
    with open(os.path.join(out_dir, fname), "w") as fh:
        json.dump(out, fh, indent=1)
    print(fname, f"{(rd + wr) / 1e9:.2f} GB per forward, x{(rd + wr) / algo:.2f} algorithmic")
