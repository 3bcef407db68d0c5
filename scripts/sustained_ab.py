#!/usr/bin/env python3
"""Sustained-load A/B of the 160x160 (16x16x32 MFMA) kernel vs the 128x128 (32x32x16 MFMA) kernel: each arm runs
~2 s back-to-back (the chip settles at its power-limited clock), arms alternate, the last half of each burst is
timed.  Short interleaved bursts (scripts/tile16_bench.py) run at boost clocks and can mislead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()

def burst(fn, seconds=2.0):
    fn(); torch.cuda.synchronize()
    t_end = time.time() + seconds / 2
    while time.time() < t_end:                     # settle
        for _ in range(50): fn()
        torch.cuda.synchronize()
    n, e0, e1 = 0, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t_end = time.time() + seconds / 2
    while time.time() < t_end:
        for _ in range(50): fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def ab(fn, flops, tag):
    out = {"0": [], "2": []}
    for _ in range(2):
        for mode in ("0", "2"):
            os.environ["TC_GEMM_TILE16"] = mode
            out[mode].append(burst(fn))
    t0, t2 = sum(out["0"]) / 2, sum(out["2"]) / 2
    print(f"{tag:28s} sustained: 128-tile {t0*1e3:7.1f} us {flops/t0/1e9:7.1f} TF/s | tile16 {t2*1e3:7.1f} us {flops/t2/1e9:7.1f} TF/s | x{t0/t2:5.2f}", flush=True)

def conv(frames, h, w, cin, cout, tag):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    wt = (torch.randn(cout, 9 * cin, device=dev) * (9 * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ab(lambda: hip.gemm(x, wt, b, conv=geom), 2.0 * frames * h * w * cout * 9 * cin, f"conv3x3 {tag} {cin}->{cout}")

def lin(m, n, k, tag):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    ab(lambda: hip.gemm(a, w), 2.0 * m * n * k, f"linear {tag}")

conv(32, 40, 64, 320, 320, "L0"); conv(32, 20, 32, 640, 640, "L1"); lin(81920, 320, 1280, "L0 ff2"); lin(20480, 640, 2560, "L1 ff2")
