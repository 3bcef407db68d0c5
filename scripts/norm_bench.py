#!/usr/bin/env python3
"""Microbenchmark of GroupNorm(+SiLU) and LayerNorm at the shapes of the UNet (B=2) and decoder: time per
call and the HBM rate implied by the algorithmic bytes (GroupNorm: x read twice + y written = 6 B/element;
LayerNorm: 4 B/element)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
from gemm_bench import timeit
hip = ops.backend(); dev = "cuda"; BF = torch.bfloat16
print("TC_GN_BLOCKS", os.environ.get("TC_GN_BLOCKS"))


def gn(samples, rows, c, tag):
    """TB/s on the ALGORITHMIC bytes (2 B read + 2 B written per element), the figure bench.py's roofline_hbm quotes.  (Rounds 5 / 6
    compared this routing with a cooperative single-launch kernel and with 768-thread one-pass blocks here; both lost and are gone:
    profiles/r05_gn_coop_bench.txt, r06_gn_onepass768_bench.txt.)"""
    x = torch.randn(samples * rows, c, device=dev).to(BF)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ms = timeit(lambda: hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True))
    print(f"groupnorm {tag:18s} s={samples:3d} rows={rows:7d} c={c:5d} | {ms*1e3:7.1f} us {4.0*x.numel()/ms/1e9:5.2f} TB/s")


def ln(rows, c, tag):
    x = torch.randn(rows, c, device=dev).to(BF)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ms = timeit(lambda: hip.layernorm(x, g, b))
    print(f"layernorm {tag:22s} rows={rows:7d} c={c:5d}  {ms*1e3:8.1f} us  {4.0*x.numel()/ms/1e9:7.2f} TB/s")


gn(2, 40960, 320, "L0 clip-wide"); gn(32, 2560, 320, "L0 per-frame"); gn(32, 2560, 640, "L0 per-frame 640"); gn(32, 2560, 960, "L0 per-frame 960")
gn(2, 10240, 640, "L1 clip-wide"); gn(32, 640, 640, "L1 per-frame"); gn(32, 640, 1920, "L1 per-frame 1920"); gn(2, 2560, 1280, "L2 clip-wide")
gn(32, 160, 1280, "L2 per-frame"); gn(2, 640, 1280, "L3 clip-wide"); gn(32, 40, 1280, "L3 per-frame")
gn(16, 163840, 128, "dec L0 per-frame"); gn(1, 2621440, 128, "dec L0 clip-wide"); gn(16, 40960, 256, "dec L1 per-frame")
gn(16, 10240, 512, "dec L2 per-frame")
ln(81920, 320, "L0"); ln(20480, 640, "L1"); ln(5120, 1280, "L2"); ln(1280, 1280, "L3")
