#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the GEMM kernels: representative linear / convolution shapes of one
B=2 UNet forward (random bf16), two launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE
hip = ops.backend(); dev = "cuda"; BF = torch.bfloat16
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(BF)
for m, n, k, act, res in [(81920, 320, 320, ACT_NONE, True), (81920, 2560, 320, ACT_GEGLU, False), (81920, 320, 1280, ACT_NONE, True),
                          (20480, 1920, 640, ACT_NONE, False), (5120, 1280, 5120, ACT_NONE, True)]:
    a, w, b = rnd(m, k), rnd(n, k, scale=k ** -0.5), torch.randn(n, device=dev)
    r = rnd(m, n // 2 if act == ACT_GEGLU else n) if res else None
    for _ in range(2):
        hip.gemm(a, w, b, act=act, residual=r)
    torch.cuda.synchronize()
for frames, h, w_, cin, cout in [(32, 40, 64, 320, 320), (32, 20, 32, 640, 640), (32, 5, 8, 1280, 1280)]:
    x, wt, b = rnd(frames * h * w_, cin), rnd(cout, 9 * cin, scale=(9 * cin) ** -0.5), torch.randn(cout, device=dev)
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
    for _ in range(2):
        hip.gemm(x, wt, b, conv=geom)
    torch.cuda.synchronize()
