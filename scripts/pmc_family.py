#!/usr/bin/env python3
"""Per kernel family of a guided UNet forward (scripts/pmc_family.sh): share of time, matrix-pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the normalisation of DESIGN.md 5.6 (4)), fraction of wave
cycles spent waiting, and L1 -> L2 read requests (TCP_TCC_READ_REQ, 128 B each: 67.1 M requests for the 8.6 GB of a 256x256 tiling of 8192^3) per second.
usage: pmc_family.py pass_a.db pass_c.db"""
import sqlite3
import sys

FAMS = [("gemm_kernel<0", "linear, 128^2 / 64^2 tiles"), ("gemm_kernel<1", "3x3 conv, 128^2 tiles / K slices"),
        ("gemm_kernel<2", "temporal conv, 128^2 / 64^2 tiles"), ("gemm16_kernel", "160^2 tiles"), ("conv_halo_kernel", "3x3 conv, halo patches"),
        ("gemm8_kernel", "8-wave 256^2"), ("gemm_wide_kernel", "256-row tiles"), ("gemm_ws_kernel", "K = 320 weight-stationary"),
        ("ff_fused_kernel", "level-0 feed-forward, one launch"), ("tb_fused_kernel", "level-0 temporal attention, one launch"),
        ("qkv_attn_kernel", "temporal qkv projection + attention, one launch"),
        ("attn_d64", "attention d = 64"), ("attn_temporal", "temporal attention"), ("gn_", "GroupNorm"), ("layernorm", "LayerNorm"),
        ("splitk_reduce", "split-K reduction")]


def fam_of(name):
    for key, label in FAMS:
        if key in name:
            return label
    return "other"


def load(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counter" in t.lower()][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "kernel" in x.lower()][0]
    ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x.lower() and "name" in x.lower()][0]
    vcol = "value" if "value" in cols else [x for x in cols if "value" in x.lower()][0]
    ctr = {}
    for k, cn, v in c.execute(f"select {kcol}, {ccol}, {vcol} from {view}"):
        f = ctr.setdefault(fam_of(k), {})
        f[cn] = f.get(cn, 0.0) + float(v)
    dur = {}
    for k, d in c.execute("select name, duration from kernels"):
        f = fam_of(k)
        dur[f] = dur.get(f, 0.0) + float(d)
    return ctr, dur


a, dur_a = load(sys.argv[1])
cc, dur_c = load(sys.argv[2])
tot = sum(v for k, v in dur_a.items() if k != "other")
print("# one pass = 2 eager guided (B = 2) UNet forwards under rocprofv3 --pmc; durations are those of the counter pass itself")
print(f"{'family':44s} {'time %':>7s} {'MFMA busy':>10s} {'waiting':>8s} {'MFMA inst / VALU inst':>22s} {'L1->L2 read req TB/s':>21s} {'L2 hit':>7s}")
for label in [l for _, l in FAMS]:                      # ("other" = the model-build kernels of the process)
    if label not in dur_a:
        continue
    x, y = a.get(label, {}), cc.get(label, {})
    gui = y.get("GRBM_GUI_ACTIVE", 0.0)
    busy = x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else float("nan")
    wait = x.get("SQ_WAIT_ANY", 0.0) / x["SQ_WAVE_CYCLES"] if x.get("SQ_WAVE_CYCLES") else float("nan")
    ratio = x.get("SQ_INSTS_MFMA", 0.0) / x["SQ_INSTS_VALU"] if x.get("SQ_INSTS_VALU") else float("nan")
    req = y.get("TCP_TCC_READ_REQ_sum", 0.0) * 128.0 / (dur_c.get(label, 0.0) * 1e-9) / 1e12 if dur_c.get(label) else float("nan")
    hm = y.get("TCC_HIT_sum", 0.0) + y.get("TCC_MISS_sum", 0.0)
    hit = y.get("TCC_HIT_sum", 0.0) / hm if hm else float("nan")
    print(f"{label:44s} {100 * dur_a[label] / tot:7.1f} {busy:10.3f} {wait:8.3f} {ratio:22.3f} {req:21.2f} {hit:7.3f}")
print(f"# all kernels: {tot / 2e6:.2f} ms per forward in the counter pass")
