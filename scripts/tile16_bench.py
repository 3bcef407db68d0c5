#!/usr/bin/env python3
"""A/B of the 160x160-tile kernel (gemm16.hip) against the 128x128 / 256-row kernels on the level-0 / level-1
shapes of a B=2 UNet forward: TC_GEMM_TILE16 = 0 (off) vs 2 (forced), interleaved in one process."""
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

def ab(fn, flops, tag):
    res = {}
    for rnd in range(2):
        for mode in ("0", "2"):
            os.environ["TC_GEMM_TILE16"] = mode
            res.setdefault(mode, []).append(timeit(fn))
    t0, t2 = min(res["0"]), min(res["2"])
    print(f"{tag:30s} off {t0*1e3:8.1f} us {flops/t0/1e9:7.1f} TF/s | tile16 {t2*1e3:8.1f} us {flops/t2/1e9:7.1f} TF/s | x{t0/t2:5.2f}")

def lin(m, n, k, tag, res=True):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev).to(BF) if res else None
    ab(lambda: hip.gemm(a, w, b, residual=r), 2.0 * m * n * k, f"linear {tag} {m}x{n}x{k}")

def conv(frames, h, w, cin, cout, tag, t3=False):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(BF)
    b = torch.randn(cout, device=dev)
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ab(lambda: hip.gemm(x, wt, b, conv=geom), 2.0 * frames * h * w * cout * taps * cin, f"{'convT3' if t3 else 'conv3x3'} {tag} {cin}->{cout}")

lin(81920, 320, 320, "L0 proj"); lin(81920, 960, 320, "L0 qkv", res=False); lin(81920, 320, 1280, "L0 ff2")
conv(32, 40, 64, 320, 320, "L0"); conv(32, 40, 64, 640, 320, "L0"); conv(32, 40, 64, 960, 320, "L0")
conv(32, 40, 64, 320, 320, "L0", t3=True)
lin(20480, 640, 640, "L1 proj"); lin(20480, 1920, 640, "L1 qkv", res=False); lin(20480, 640, 2560, "L1 ff2")
conv(32, 20, 32, 640, 640, "L1"); conv(32, 20, 32, 1280, 640, "L1"); conv(32, 20, 32, 320, 640, "L1")
conv(32, 20, 32, 640, 640, "L1", t3=True)
lin(5120, 1280, 1280, "L2 proj"); conv(32, 10, 16, 1280, 1280, "L2")
lin(4096, 4000, 4096, "square-ish 4k (N=4000)", res=False)
