#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: one launch of each dominant GEMM shape of a B=2 UNet
forward (random bf16 data), preceded by a cache-flushing fill so that FETCH_SIZE reflects HBM, not
Infinity-Cache hits from the previous launch.  Prints the algorithmic bytes of every launch
(A read once + W read once + C written once + residual read once)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)          # 512 MiB > 256 MiB Infinity Cache
shapes = [("L0_proj", 81920, 320, 320, ACT_NONE, True), ("L0_qkv", 81920, 960, 320, ACT_NONE, False),
          ("L0_geglu", 81920, 2560, 320, ACT_GEGLU, False), ("L0_ff2", 81920, 320, 1280, ACT_NONE, True),
          ("L1_geglu", 20480, 5120, 640, ACT_GEGLU, False), ("L2_ff2", 5120, 1280, 5120, ACT_NONE, True)]
out = []
for tag, m, n, k, act, res in shapes:
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev).to(BF) if res else None
    n_out = n // 2 if act == ACT_GEGLU else n
    hip.gemm(a, w, b, act=act, residual=r); torch.cuda.synchronize()        # warm code
    flush.fill_(1.0); torch.cuda.synchronize()
    hip.gemm(a, w, b, act=act, residual=r); torch.cuda.synchronize()        # the measured launch (cold caches)
    alg = 2 * (m * k + n * k + m * n_out + (m * n_out if res else 0))
    out.append(dict(tag=tag, m=m, n=n, k=k, algorithmic_bytes=alg, flops=2.0 * m * n * k))
print(json.dumps(out))
