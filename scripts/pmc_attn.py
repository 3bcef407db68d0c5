#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the attention kernels: the UNet's level-0/1 self-attention, the text
and image cross-attention, the temporal attention and the decoder's reference attention shape (random bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
hip = ops.backend(); dev = "cuda"; BF = torch.bfloat16
def rnd(*s): return torch.randn(*s, device=dev).to(BF)
cases = [(32, 5, 2560, 2560, 1), (32, 10, 640, 640, 1), (32, 5, 2560, 77, 16), (32, 5, 2560, 16, 1)]
for (b, h, lq, lk, div) in cases:
    kvb = (b + div - 1) // div
    q, k, v = rnd(b * lq, h * 64), rnd(kvb * lk, h * 64), rnd(kvb * lk, h * 64)
    for _ in range(2):
        hip.attention(q, k, v, batch=b, heads=h, lq=lq, lk=lk, kv_bdiv=div)
    torch.cuda.synchronize()
qkv = rnd(2 * 16 * 2560, 3 * 320)
for _ in range(2):
    hip.attention_temporal(qkv, b=2, t=16, hw=2560, heads=5)
torch.cuda.synchronize()
