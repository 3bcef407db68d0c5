#!/usr/bin/env python3
"""How much do tile quantisation (blocks vs the 512 resident slots of the 128x128 kernel) and N padding cost?
Same kernel, shapes chosen so that only the tile count / padding changes.  TF/s per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()

def timeit(fn, iters=20, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]

def lin(m, n, k, tag):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    ms = timeit(lambda: hip.gemm(a, w))
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    print(f"{tag:34s} M={m:6d} N={n:5d} K={k:5d} tiles128={tiles:5d} ({tiles/512:5.2f} rounds) {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF/s")

print("# N padding at L0 (M=81920): N=320 computes 3 N-tiles like N=384")
for k in (320, 1280, 2880):
    lin(81920, 320, k, "L0 N=320"); lin(81920, 384, k, "L0 N=384 (no padding)"); lin(81920, 256, k, "L0 N=256 (2 tiles)")
print("# quantisation at L1/L2: same shape family, tile count at / off a multiple of 512")
for k in (640, 2560, 5760):
    lin(20480, 640, k, "L1 800 tiles"); lin(16384, 512, k, "512 tiles (1.0 round)"); lin(32768, 512, k, "1024 tiles (2.0 rounds)")
for k in (1280, 5120, 11520):
    lin(5120, 1280, k, "L2 400 tiles"); lin(8192, 1024, k, "512 tiles (1.0 round)"); lin(4096, 1024, k, "256 tiles (0.5 round)")
