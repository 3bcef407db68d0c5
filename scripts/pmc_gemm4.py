#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: the square 8192^3 bf16 product on (i) the 8-wave kernel (gemm8.hip, default routing),
(ii) the four-wave kernel (gemm4.hip, TC_GEMM4=2), (iii) hipBLASLt's own 256x256x64 kernel (torch F.linear -- script only); three
launches each.  The question the counters answer: with the same tile, the same bytes and (gemm4) the same per-wave instruction
mix, where do this library's kernels spend the cycles the library's does not?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tooncrafter_amd.ops import HipOps
hip = HipOps(); dev = "cuda"; BF = torch.bfloat16
m = n = k = 8192
a = torch.randn(m, k, device=dev).to(BF)
w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
for mode in ("0", "2"):
    os.environ["TC_GEMM4"] = mode
    for _ in range(3):
        hip.gemm(a, w)
    torch.cuda.synchronize()
for _ in range(3):
    F.linear(a, w)
torch.cuda.synchronize()
