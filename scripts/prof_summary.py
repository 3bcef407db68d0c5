#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace of `bench.py`: per-kernel calls / total / average, per-(kernel, grid)
breakdown, per DDIM step shares, and the GEMM-family roofline fraction recomputed from THIS file alone.

usage: prof_summary.py <results.db> [top_n] [bench_stdout_log] [ddim_steps]

What is in the tables: only the PRODUCT's kernels inside the product window -- from the first to the last
kernel of libtooncrafter_hip.so (their names carry the library's anonymous namespace) -- plus the few ATen kernels the sampler
itself launches inside that window (`randn`, `full`).  Left out, and reported in one line each so that nothing disappears
silently: the hipBLASLt calibration matmuls of bench.py's `lease_calibration` (`Cijk_*`: not on the product path) and every
kernel before / after the window (model build: weight synthesis and packing; the after-the-fact extras).  With
`bench_stdout_log` (the file bench.py's JSON line went to) the lease calibration of the SAME run is printed beside the tables;
`ddim_steps` (default 10: the trace command of scripts/gpu_r6_*.sh) sizes the algorithmic FLOPs of the run.  GEMM family:
sum of 2 M N K over its launches = 22.661 TFLOP per guided (B = 2) forward (402 launches; the layers in front of the first
cross-attention run once for both passes) + 57.32 TFLOP for the 16-frame and 14-frame decodes (158 launches) -- the figures
bench.py's probe counts launch by launch and prints as `roofline.unet.algorithmic_tflop_per_unet_fwd_b2` /
`roofline.decoder.algorithmic_tflop_16f_plus_14f`; they are properties of the model, not of a run.  Whole clip: the
reference modules' work, 2 x 12.603 TFLOP per guided forward + 37.875 + 33.148 (1 MAC = 2 FLOP).
"""
import json
import re
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
bench_log = sys.argv[3] if len(sys.argv) > 3 else None
ddim_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
PEAK_TF = 2500.0
TFLOP_FWD_B1, TFLOP_DEC16, TFLOP_DEC14 = 12.603, 37.875, 33.148
GEMM_TFLOP_FWD_B2, GEMM_TFLOP_DECODES = 22.661, 57.32

c = sqlite3.connect(db)
# libtooncrafter_hip.so keeps every kernel in an anonymous namespace; so do some ATen kernels (`at::native::(anonymous
# namespace)::distribution_elementwise_...`, `CatArrayBatchedCopy`): those are NOT the product's
is_product = lambda n: ("anonymous namespace" in n or "_GLOBAL__N_" in n) and "at::native" not in n and "at::cuda" not in n
is_calib = lambda n: n.startswith("Cijk_")
# the GEMM family of bench.py's `roofline`: every tc_gemm_bf16 kernel, the one-launch operators that contain projections
GEMM_FAMILY = re.compile(r"gemm\w*_kernel|conv_halo_kernel|ff_fused_kernel|tb_fused_kernel|qkv_attn_kernel|splitk_reduce_kernel")

allk = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
prod = [k for k in allk if is_product(k[0])]
if not prod:
    sys.exit("no product kernels in the trace")
w0, w1 = prod[0][1], max(k[2] for k in prod)
inside = [k for k in allk if w0 <= k[1] <= w1 and not is_calib(k[0])]
calib = [k for k in allk if is_calib(k[0])]
outside = [k for k in allk if (k[1] < w0 or k[1] > w1) and not is_calib(k[0])]
dur = lambda ks: sum(k[2] - k[1] for k in ks)

if bench_log:
    try:
        for line in open(bench_log):
            if line.startswith("{") and "lease_calibration" in line:
                d = json.loads(line)
                lc = d["lease_calibration"]
                mm = [lc[k].get("matmul_8192_bf16_tflops") for k in ("before_timed_region", "after_timed_region") if lc.get(k)]
                cp = [lc[k].get("copy_1gib_gbs") for k in ("before_timed_region", "after_timed_region") if lc.get(k)]
                print(f"# lease calibration of this run (under the profiler): hipBLASLt 8192^3 bf16 {mm} TF/s, 1 GiB copy {cp} GB/s "
                      f"(reference lease: 1200 TF/s, 5200 GB/s); value {d.get('value')} {d.get('unit')} at ddim_steps {ddim_steps}")
    except (OSError, ValueError, KeyError) as e:
        print(f"# lease calibration: not available ({type(e).__name__}: {e})")
print(f"# product window: {len(inside)} dispatches, busy {dur(inside) / 1e6:.1f} ms, span {(w1 - w0) / 1e6:.1f} ms")
print(f"# left out of the tables: {len(calib)} calibration matmuls (Cijk_*) {dur(calib) / 1e6:.1f} ms; "
      f"{len(outside)} kernels outside the window (model build, extras) {dur(outside) / 1e6:.1f} ms")

agg = {}
for k in inside:
    e = agg.setdefault(k[0], [0, 0])
    e[0] += 1
    e[1] += k[2] - k[1]
tot = dur(inside)
print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for name, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{name[:80]:80s} {n:7d} {d / 1e6:10.2f} {d / n / 1e3:9.1f} {100 * d / tot:6.1f}")

fam = [k for k in inside if GEMM_FAMILY.search(k[0])]
n_marks = sum(1 for k in inside if "ddim_apply_kernel" in k[0])
clips = max(1, round(n_marks / ddim_steps))                       # a warm-up clip in the trace doubles the window's work
tflop = clips * (ddim_steps * GEMM_TFLOP_FWD_B2 + GEMM_TFLOP_DECODES)
clip_tflop = clips * (ddim_steps * 2 * TFLOP_FWD_B1 + TFLOP_DEC16 + TFLOP_DEC14)
print(f"\n# GEMM family (gemm* / conv_halo / ff_fused / tb_fused / qkv_attn / splitk_reduce): {len(fam)} launches, {dur(fam) / 1e6:.1f} ms "
      f"for {tflop:.1f} algorithmic TFLOP ({clips} x ({ddim_steps} guided forwards + 2 decodes)) = {tflop / (dur(fam) / 1e9):.0f} TF/s = "
      f"{tflop / (dur(fam) / 1e9) / PEAK_TF:.3f} of the {PEAK_TF:.0f} TF/s bf16 MFMA peak (a LOWER bound: the window also holds the "
      f"un-captured passes in front of the hipGraph captures and the encoder pass behind the clip; the per-clip composition at the end of this file is the comparable figure)")
print(f"# whole window: {clip_tflop:.1f} TFLOP of reference-module work over {tot / 1e6:.1f} ms of kernel-busy time = "
      f"{clip_tflop / (tot / 1e9):.0f} TF/s = {clip_tflop / (tot / 1e9) / PEAK_TF:.3f} of peak")

print("\n# per (kernel, grid) -- top 40 by total time")
g = {}
for k in inside:
    e = g.setdefault((k[0], k[3], k[4], k[5]), [0, 0])
    e[0] += 1
    e[1] += k[2] - k[1]
for (name, gx, gy, gz), (n, d) in sorted(g.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{name[:60]:60s} grid({gx},{gy},{gz}) calls {n:6d} total {d / 1e6:9.2f} ms avg {d / n / 1e3:9.1f} us")

# ---- per DDIM step (between consecutive ddim_apply_kernel launches = one guided UNet forward + the step): busy time,
# idle gaps between kernels, and the per-kernel shares averaged over the steps
marks = [k[1] for k in inside if "ddim_apply_kernel" in k[0]]
# With a warm-up clip in the trace (bench.py --warmup 1 --steps 1: the hipGraphs of the forward and of both decodes are captured
# there) only the LAST clip is analysed: its ddim_steps marks and the decodes behind the last of them.  Without one, the first
# interval (eager pass + capture) is skipped and the decodes carry their first-use passes (then flagged below).
warm = len(marks) >= 2 * ddim_steps
if warm:
    marks = marks[-ddim_steps:]
if len(marks) >= 3:
    spans = []
    sagg = {}
    for a, b in (zip(marks[:-1], marks[1:]) if warm else zip(marks[1:-1], marks[2:])):
        ks = [k for k in inside if a <= k[1] < b]
        busy = dur(ks)
        gaps = sum(max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1))
        spans.append((b - a, busy, gaps, len(ks)))
        for k in ks:
            e = sagg.setdefault(k[0], [0, 0])
            e[0] += 1
            e[1] += k[2] - k[1]
    n = len(spans)
    fam_ms = sum(d for name, (_, d) in sagg.items() if GEMM_FAMILY.search(name)) / n / 1e6
    print(f"\n# per DDIM step ({n} steps between ddim_apply_kernel launches{' of the LAST clip (graphs captured in the warm-up clip)' if warm else ''}): span {sum(s[0] for s in spans) / n / 1e6:.2f} ms, "
          f"kernel-busy {sum(s[1] for s in spans) / n / 1e6:.2f} ms, gaps between kernels {sum(s[2] for s in spans) / n / 1e6:.2f} ms, "
          f"{sum(s[3] for s in spans) / n:.0f} kernels; GEMM family {fam_ms:.2f} ms = {GEMM_TFLOP_FWD_B2 / (fam_ms / 1e3):.0f} TF/s "
          f"= {GEMM_TFLOP_FWD_B2 / (fam_ms / 1e3) / PEAK_TF:.3f} of peak")
    for name, (cnt, d) in sorted(sagg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:80]:80s} {cnt / n:7.1f} {d / n / 1e6:10.3f} ms {d / cnt / 1e3:9.1f} us {100 * d / sum(s[1] for s in spans):6.1f}")
    # ---- the clip's composition, as bench.py's `roofline` composes it: 50 x (one guided forward) + the two decodes.  The decodes are
    # the product kernels behind the LAST ddim_apply launch.  (The window-wide figure above also holds the un-captured forward that
    # precedes the hipGraph capture -- one forward's time more than `ddim_steps` forwards' FLOPs -- so THIS is the figure to compare.)
    # (... and in front of the clip's last kernel, tc_video_to_u8: bench.py times the first-stage ENCODER right behind the timed region)
    ends = [k[1] for k in inside if k[1] > marks[-1] and "video_to_u8" in k[0]]
    dec_end = ends[0] if ends else w1
    dec = [k for k in inside if marks[-1] < k[1] <= dec_end and GEMM_FAMILY.search(k[0])]
    if dec:
        dec_ms = dur(dec) / 1e6
        clip_ms = 50 * fam_ms + dec_ms
        clip_tf = 50 * GEMM_TFLOP_FWD_B2 + GEMM_TFLOP_DECODES
        if not warm:
            print("\n# (no warm-up clip in this trace: the decodes below include their un-captured first-use passes)")
        print(f"\n# GEMM family, decodes (16 + 14 frames, {len(dec)} launches behind the last DDIM step): {dec_ms:.1f} ms = "
              f"{GEMM_TFLOP_DECODES / (dec_ms / 1e3):.0f} TF/s = {GEMM_TFLOP_DECODES / (dec_ms / 1e3) / PEAK_TF:.3f} of peak")
        print(f"# GEMM family, one clip = 50 x the step above + the decodes: {clip_tf:.1f} TFLOP in {clip_ms:.1f} ms = "
              f"{clip_tf / (clip_ms / 1e3):.0f} TF/s = {clip_tf / (clip_ms / 1e3) / PEAK_TF:.3f} of peak   <- bench.py roofline.frac, recomputed from this trace")
