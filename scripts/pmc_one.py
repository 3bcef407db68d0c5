#!/usr/bin/env python3
"""A few launches of selected GEMM shapes for PMC passes (counters are per dispatch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
hip = ops.backend(); dev = "cuda"; BF = torch.bfloat16
for (m, n, k) in ((4096, 4096, 4096), (81920, 320, 320), (81920, 960, 320)):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    for _ in range(3):
        hip.gemm(a, w)
    torch.cuda.synchronize()
x = torch.randn(32 * 2560, 320, device=dev).to(BF); wt = (torch.randn(320, 2880, device=dev) * 0.02).to(BF)
geom = dict(kind="3x3", frames=32, cin=320, h_in=40, w_in=64, h_out=40, w_out=64, stride=1, upsample=False)
for _ in range(3):
    hip.gemm(x, wt, conv=geom)
torch.cuda.synchronize()
